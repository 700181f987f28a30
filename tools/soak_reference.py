"""Soak: many seeded worlds / image pairs, product (HIP library) vs the reference's own compiled sources (oracle/_ref).  Run on a GPU box:
    python tools/soak_reference.py [n_world_seeds [n_stereo_pairs]]
Prints one line per family and exits non-zero on the first difference."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                                     # noqa: E402
from orb_slam3_detailed_comments_amd import _lib, ORBextractor, ComputeStereoMatches, synth   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else n
# even seeds: stand-in Frame / KeyFrame / MapPoint world; odd seeds: the reference's own classes (Frame.cc, KeyFrame.cc, MapPoint.cc compiled in place)
WORLDS = [(os.path.join(ROOT, "oracle", "_ref", "libmw_ref.so"), os.path.join(ROOT, "oracle", "_ref", "libmw_facade.so")),
          (os.path.join(ROOT, "oracle", "_ref", "libmw_ref_full.so"), os.path.join(ROOT, "oracle", "_ref", "libmw_facade_full.so"))]
RUN = os.path.join(ROOT, "tests", "matcher_world.py")
tmp = tempfile.mkdtemp()
bad = 0
for seed in range(1000, 1000 + n):
    REF, FAC = WORLDS[seed & 1]
    for variant in ("base", "dense", "hard", "rig"):
        a = os.path.join(tmp, "a.npz"); b = os.path.join(tmp, "b.npz")
        subprocess.run([sys.executable, RUN, REF, "", str(seed), variant, a], check=True)
        subprocess.run([sys.executable, RUN, FAC, (os.environ.get("ORBX_SOAK_LIB") or _lib.HIP_LIB_PATH), str(seed), variant, b], check=True)
        A, B = np.load(a), np.load(b)
        for k in A.files:
            if k != "flavour" and not np.array_equal(A[k], B[k]):
                print("DIFF world seed %d %s: %s" % (seed, variant, k)); bad += 1
print("matcher worlds: %d seeds x 4 variants, %d differences" % (n, bad))

LIB_PATH = os.environ.get("ORBX_SOAK_LIB") or _lib.HIP_LIB_PATH      # ORBX_SOAK_LIB=tests/emu/liborbx_emu.so: the same soak on the CPU emulator build
lib = _lib.OrbxLib(LIB_PATH) if os.environ.get("ORBX_SOAK_LIB") else _lib.load_hip()
FX = 458.654; BF = FX * 0.110074
bad2 = 0
exs = {}
for seed in range(2000, 2000 + npairs):
    w, h = [(752, 480), (640, 480), (376, 240), (376, 240)][seed % 4]
    nf = [1200, 1000, 500, 500][seed % 4]
    kind = seed % 7
    if kind == 5:                                               # camera-like statistics: some cells need the second FAST run at minThFAST
        L, R = synth.natural_stereo_pair(w, h, seed=seed)
    elif kind == 6:                                             # weak dots / tied strong pairs per cell: the threshold fallback and the strict-maximum rule
        L = synth.threshold_blocks(w, h, seed=seed); R = np.roll(L, -(3 + seed % 20), axis=1)
    elif kind == 3:
        L = synth.pink_noise(w, h, seed=seed); R = np.roll(L, -7, axis=1)
    elif kind == 4:                                             # exact descriptor copies along the rows: the tie rule of the row search
        L, R = synth.periodic_stereo_pair(w, h, seed=seed, period=[(48, 240), (32, 120), (64, 480)][seed % 3], disparity=5 + seed % 40)
    else:
        L, R = synth.stereo_pair(w, h, seed=seed, nrect=[2000, 800, 3000][kind])
    F = ol.ReferenceFrame(L, R, nf, fx=FX, bf=BF)
    ex = exs.get(nf) or exs.setdefault(nf, ORBextractor(nf, 1.2, 8, 20, 7, lib=lib))
    (_, kL, dL), (_, kR, dR) = ex.extract_batch(np.stack([L, R]))
    u, d, m = ComputeStereoMatches(ex, ex, BF, F.mb, 0, 1, 1)
    ok = (kL.tobytes() == F.keys.tobytes() and dL.tobytes() == F.desc.tobytes() and kR.tobytes() == F.keys_right.tobytes() and dR.tobytes() == F.desc_right.tobytes()
          and u[0, :F.N].tobytes() == F.u_right.tobytes() and d[0, :F.N].tobytes() == F.depth.tobytes())
    if not ok:
        print("DIFF frame seed %d (%dx%d)" % (seed, w, h)); bad2 += 1
print("stereo frames: %d pairs, %d differences" % (npairs, bad2))

# Batch-size dependent code paths (quadtree threads per tree, keypoints per wave in the describe kernel, ...): batches of 40 and 96 images of one
# size must give, image by image, what pairs of two give (which the loop above has compared with the reference).
bad3 = 0; nb = 0
for rep in range(max(1, npairs // 100)):
    w, h, nf = [(752, 480, 1200), (640, 480, 1000)][rep % 2]
    B = [40, 96][rep % 2]
    imgs = []
    for i in range(B // 2):
        L, R = synth.stereo_pair(w, h, seed=9000 + 100 * rep + i, nrect=[2000, 800, 3000][i % 3]); imgs += [L, R]
    ex = exs.get(nf) or exs.setdefault(nf, ORBextractor(nf, 1.2, 8, 20, 7, lib=lib))
    big = ex.extract_batch(np.stack(imgs))
    for i in range(0, B, 2):
        small = ex.extract_batch(np.stack(imgs[i:i + 2]))
        for a2, b2 in zip(big[i:i + 2], small):
            nb += 1
            if not (a2[0] == b2[0] and a2[1].tobytes() == b2[1].tobytes() and a2[2].tobytes() == b2[2].tobytes()):
                print("DIFF batch rep %d image %d" % (rep, i)); bad3 += 1
print("batched vs pairwise extraction: %d images, %d differences" % (nb, bad3))
sys.exit(1 if bad or bad2 or bad3 else 0)
