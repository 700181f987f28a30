"""Shared parity cases: (name, image factory, nfeatures, lapping area) — SURVEY.md §8d configs on synthetic inputs."""
from orb_slam3_detailed_comments_amd import synth

FULL_CASES = [
    ("euroc_752x480_n1200", lambda: synth.corner_field(seed=0), 1200, (0, 0)),
    ("euroc_mono_n1000_lap_all", lambda: synth.corner_field(seed=1), 1000, (0, 1000)),
    ("low_texture", lambda: synth.corner_field(seed=2, contrast_div=6.0), 1200, (0, 0)),
    ("sparse_corners", lambda: synth.sparse_corners(seed=0), 1200, (0, 0)),
    ("uniform_noise_partial_lap", lambda: synth.uniform_noise(seed=0), 1200, (100, 400)),
    ("tumvi_512_n1500_lap", lambda: synth.corner_field(512, 512, seed=3), 1500, (0, 511)),
    ("tum_640x480_n1000", lambda: synth.corner_field(640, 480, seed=4), 1000, (0, 0)),
    ("pink_noise_752x480_n1200", lambda: synth.pink_noise(seed=6), 1200, (0, 0)),
    ("threshold_fallback_752x480", lambda: synth.threshold_blocks(752, 480, seed=7), 1200, (0, 0)),
    ("euroc_mono_600x350_n5000_init", lambda: synth.corner_field(600, 350, seed=5), 5000, (0, 1000)),
]

LARGE_CASES = [   # sizes outside BASELINE.json that real rigs use (KITTI, HD): GPU parity only
    ("kitti_1241x376_n2000", lambda: synth.corner_field(1241, 376, seed=20, nrect=3900), 2000, (0, 0)),
    ("hd_1920x1080_n3000", lambda: synth.corner_field(1920, 1080, seed=21, nrect=17000), 3000, (0, 0)),
    # 226 k FAST candidates on level 0: the quadtree's 32-bit sort counters (levels with >= 65 536 keys) and its workgroup-wide partitions
    ("noise_1920x1200_n3000", lambda: synth.uniform_noise(1920, 1200, seed=22), 3000, (0, 0)),
    ("noise_1600x1200_n1500", lambda: synth.uniform_noise(1600, 1200, seed=23), 1500, (0, 1000)),
    # the largest image the library takes (keypoint coordinates are packed in 12 bits per axis on the device: 4095 + the 2 x 16-px border) - 17 Mpixel,
    # 10 148 FAST cells on level 0; with noise every cell overflows its list and the quadtree starts from 2.6 M candidates
    ("max_4127x4127_n5000", lambda: synth.corner_field(4127, 4127, seed=5, nrect=int(3000 * 4127 * 4127 / (752 * 480))), 5000, (0, 0)),
    ("max_noise_4127x4127_n6000", lambda: synth.uniform_noise(4127, 4127, seed=6), 6000, (0, 0)),     # (about 6 100 is what the quadtree's LDS takes at this size)
]

SMALL_CASES = [
    ("small_376x240_n500", lambda: synth.corner_field(376, 240, seed=10, nrect=800), 500, (0, 0)),
    ("small_lap_partial", lambda: synth.corner_field(376, 240, seed=11, nrect=800), 500, (100, 250)),
    ("small_noise", lambda: synth.uniform_noise(320, 256, seed=12), 400, (0, 1000)),
    ("small_sparse", lambda: synth.sparse_corners(376, 240, seed=13, ncorner=12), 500, (0, 0)),
    ("small_flat_empty", lambda: __import__("numpy").full((240, 376), 77, "uint8"), 500, (0, 0)),
    ("small_square_512", lambda: synth.corner_field(300, 300, seed=14, nrect=700), 300, (0, 299)),
    ("small_280x260_wide_cells", lambda: synth.corner_field(280, 260, seed=16, nrect=500), 300, (0, 0)),   # upper levels: one 60-px cell (128-byte FAST tile pitch)
    ("small_pink_noise", lambda: synth.pink_noise(376, 240, seed=17, beta=1.3), 500, (0, 0)),
    ("small_threshold_fallback", lambda: synth.threshold_blocks(376, 240, seed=18), 500, (0, 0)),   # cells that need the second FAST run at minThFAST
    ("small_wide_nini3", lambda: synth.corner_field(640, 250, seed=15, nrect=900), 600, (0, 0)),
    # the smallest images the reference can take with 8 levels at 1.2 (level 7: 67 px = one 35-px cell between the borders; one pixel less and its cell
    # count is 0 and it divides by it - the library refuses that size), square and at the smallest aspect ratio (0.5)
    ("min_239x239_noise_n1000", lambda: synth.uniform_noise(239, 239, seed=19), 1000, (0, 0)),
    ("min_477x239_aspect_n100", lambda: synth.corner_field(477, 239, seed=20, nrect=450), 100, (0, 0)),
]

EUROC_BF, EUROC_B = 458.654 * 0.110074, 0.110074   # Examples/Stereo/EuRoC.yaml:23,57 (fx * baseline, baseline)
