"""Runs the reference's own stereo Frame constructor compiled against the drop-in ORBextractor.h (oracle/_ref/libref_frame_dropin.so) on a
product library and dumps the resulting Frame; tests/test_frame_reference.py compares it with the all-reference Frame.
    python tests/frame_dropin_runner.py <orbx library> <w> <h> <seed> <nfeatures> <scale> <nlevels> <ini> <min> <gauss> <out.npz>"""
import sys

import numpy as np

import oracle_lib as ol
from orb_slam3_detailed_comments_amd import synth

orbx, w, h, seed, nf, sf, nl, ini, mn, gv, dst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9]), int(sys.argv[10]), sys.argv[11]
L = ol.dropin_frame_lib(orbx)
# the facade's Gaussian taps follow the library default (OpenCV 4.x); variant 1 is set through the environment the facade reads
left, right = (synth.stereo_pair(w, h, seed=seed, nrect=800) if w < 500 else synth.stereo_pair(w, h, seed=seed))
FX = 458.654
F = ol.ReferenceFrame(left, right, nf, sf, nl, ini, mn, gv, fx=FX, bf=FX * 0.110074, lib=L)
np.savez(dst, keys=F.keys, keys_un=F.keys_un, desc=F.desc, keys_right=F.keys_right, desc_right=F.desc_right, u_right=F.u_right, depth=F.depth,
         probe=np.array([len(F.features_in_area(300.0, 200.0, 40.0)), len(F.features_in_area(100.0, 100.0, 25.0, 1, 3))]))
