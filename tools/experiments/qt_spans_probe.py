"""Experiment (build/variants/liborbx_hip_qtspan.so, a build with two extra wall-clock stamps per quadtree workgroup): when do the 16 trees of a
stereo pair start and end inside the k_quadtree launch?  Serial profiling mode (the stamps live behind qt_prof)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam3_detailed_comments_amd import synth, _lib, ORBextractor
L, R = synth.stereo_pair(seed=100)
pair = np.stack([L, R])
for LV in (0, 2, 7):
  lib = _lib.OrbxLib(os.path.join(ROOT, 'build', 'variants', 'liborbx_hip_qtspan%d.so' % LV))
  ex = ORBextractor(1200, 1.2, 8, 20, 7, lib=lib)
  ex.profile(True, serial=True)
  dptr = ex.device_upload(pair)
  for it in range(6):
      ex.enqueue(None, (0, 0), device_ptr=dptr, shape=pair.shape); ex.sync()
      qp = np.zeros(48, np.int64); lib.L.orbx_debug_quadtree_profile(ex._h, qp.ctypes.data)
      if it < 3:
          continue
      sp = qp[16:48].reshape(2, 8, 2)
      t0 = sp[:, :, 0].min()
      print("run %d: stage ms %s" % (it, {k: round(v, 4) for k, v in ex.stage_ms().items()}))
      for b in range(2):
          print("  image %d: " % b + "  ".join("L%d %5.1f-%5.1f" % (l, (sp[b, l, 0] - t0) / 100, (sp[b, l, 1] - t0) / 100) for l in range(8)))
      print("  L%d phases: gather %.1f roots %.1f passes %.1f | final rounds %.1f (last: sort %.1f part %.1f ndiv %.1f rebuild %.1f) | select %.1f total %.1f; n=%d threads=%d passes=%d nodes=%d nexp=%d final rounds=%d" % ((LV,) + tuple((qp[i + 1] - qp[i]) / 100 for i in (0, 1, 2)) + ((qp[8] - qp[3]) / 100,) + tuple((qp[i + 1] - qp[i]) / 100 for i in (4, 5, 6, 7, 8)) + ((qp[9] - qp[0]) / 100, qp[10], qp[15], qp[13], qp[11], qp[12], qp[14])))
      print("     gather: tables %.1f | counts + scan %.1f | zero + barrier %.1f | keys %.1f" % ((qp[40] - qp[0]) / 100, (qp[41] - qp[40]) / 100, (qp[42] - qp[41]) / 100, (qp[1] - qp[42]) / 100))
