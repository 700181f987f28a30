"""Where the time of one orbm_search_local_points call goes (library built with -DORBX_TRACE_TIMING into build/variants/liborbx_hip_trace.so;
developer tool)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
import test_local_points as tlp
from orb_slam3_detailed_comments_amd import _lib, synth, views, matcher as M, ComputeStereoMatches
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
lib = _lib.OrbxLib(os.path.join(ROOT, "build", "variants", "liborbx_hip_trace.so"))
rng = np.random.default_rng(5)
Ls, Rs = synth.stereo_pair(640, 480, seed=7)
RF = ol.ReferenceFrame(Ls, Rs, 1000, fx=tlp.FX, fy=tlp.FY, cx=tlp.CX, cy=tlp.CY, bf=tlp.BF)
ex = ORBextractor(1000, 1.2, 8, 20, 7, lib=lib)
(_, kL, dL), _ = ex.extract_batch(np.stack([Ls, Rs]))
uu, _, _ = ComputeStereoMatches(ex, ex, tlp.BF, RF.mb, 0, 1, 1)
sfs = ex.GetScaleFactors()
fv = views.frame_view(kL, dL, sfs, 640, 480, u_right=uu[0, :RF.N], mbf=tlp.BF)
Rcw = tlp._rot(0.02, -0.03, 0.01); tcw = np.array([0.3, -0.1, 0.25], np.float32)
sp = tlp._scene(RF, rng, Rcw, tcw, 5000)
call = M.SearchLocalPoints(ex, fv, Rcw, tcw, (tlp.FX, tlp.FY, tlp.CX, tlp.CY), (0.0, 640.0, 0.0, 480.0), tlp.BF, sfs, *sp, 0.5, 3.0, False, 50.0, 0.8, prepared=True)
for _ in range(8):
    call()
