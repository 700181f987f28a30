"""Batches whose buffers cross 4 GB: the pyramid and its blurred copy of 4600 EuRoC images are 5.4 GB each (1 167 616 bytes per image; the 4-GB mark falls
on image 3678) - any batch offset computed in 32 bits would read or write the wrong image.  Six distinct stereo pairs are tiled over the batch; every image of the batch has to come out exactly as
the same image extracted in a batch of twelve (itself checked against the reference elsewhere), and so has every stereo pair - the later half of the batch lies
beyond the 4-GB mark.  Runs in a process of its own: a wild access would take the process down, not the test session."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RUNNER = r'''
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle_lib as ol
from orb_slam3_detailed_comments_amd import synth, _lib
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M
P, NPAIR, BF, BASE = 6, int(sys.argv[2]), 47.9, 0.11
pairs = [synth.stereo_pair(752, 480, seed=900 + p) for p in range(P)]
import os
LIB = _lib.OrbxLib(os.environ['ORBX_BIGBATCH_LIB']) if os.environ.get('ORBX_BIGBATCH_LIB') else None      # (the emulator build: checks this script on the CPU)
small = ORBextractor(1200, 1.2, 8, 20, 7, lib=LIB)
ref = small.extract_batch(np.stack([l for l, _ in pairs] + [r for _, r in pairs]))
small._lib.check(small._lib.L.orbm_stereo_match(small._h, 0, small._h, P, P, BF, BASE))
ru, rd, rn = M.StereoFetch(small, P)
ex = ORBextractor(1200, 1.2, 8, 20, 7, lib=LIB)
idx = np.arange(NPAIR) % P
imgs = np.concatenate([np.stack([pairs[p][0] for p in range(P)])[idx], np.stack([pairs[p][1] for p in range(P)])[idx]])     # lefts, then rights
res = ex.extract_batch(imgs)
bad = 0
for i in range(2 * NPAIR):
    want = ref[idx[i % NPAIR] + (P if i >= NPAIR else 0)]
    if not (res[i][0] == want[0] and ol.kps_equal(res[i][1], want[1]) and np.array_equal(res[i][2], want[2])): bad += 1
ex._lib.check(ex._lib.L.orbm_stereo_match(ex._h, 0, ex._h, NPAIR, NPAIR, BF, BASE))
u, d, n = M.StereoFetch(ex, NPAIR)
badm = 0
for i in range(NPAIR):
    N = ref[idx[i]][0]
    if not (n[i] == rn[idx[i]] and u[i, :N].tobytes() == ru[idx[i], :N].tobytes() and d[i, :N].tobytes() == rd[idx[i], :N].tobytes()): badm += 1
print("RESULT images %d wrong %d pairs %d wrong %d matches_per_pair %.1f" % (2 * NPAIR, bad, NPAIR, badm, float(n.mean())))
'''


def test_batch_buffers_beyond_4GB():
    r = subprocess.run([sys.executable, "-c", _RUNNER, ROOT, "2300"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert line[1:5] == ["images", "4600", "wrong", "0"] and line[5:9] == ["pairs", "2300", "wrong", "0"] and float(line[-1]) > 100, line
