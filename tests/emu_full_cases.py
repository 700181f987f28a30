"""Runs the 8 full-size parity cases through the SIMT emulator (slower than the pytest small cases; used while optimising)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as ol
from cases import FULL_CASES
from orb_slam3_detailed_comments_amd import _lib
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
emu = _lib.OrbxLib(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu', 'liborbx_emu.so'))
ok = True
for name, factory, nf, lap in FULL_CASES:
    img = factory()
    ex = ORBextractor(nf, 1.2, 8, 20, 7, lib=emu)
    t = time.time(); got = ex(img, None, lap); te = time.time() - t
    exp = ol.OracleExtractor(nf).extract(img, lap)
    same = got[0] == exp[0] and ol.kps_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2])
    ok &= same
    print(name, len(got[1]), 'SAME' if same else 'DIFF', '%.1fs' % te, flush=True)
print('ALL', ok)
sys.exit(0 if ok else 1)
