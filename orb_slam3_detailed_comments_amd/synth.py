"""Deterministic synthetic inputs (no dataset is available offline).

S1 "corner field": random bright/dark rectangles over a low-frequency gradient plus Gaussian noise,
which yields >= 10x the per-level feature quota in FAST candidates, like a textured EuRoC frame.
The stereo pair renders the same rectangles with a per-rectangle disparity (depth layers) and
independent noise, so the row search / SAD / parabola stages of Frame::ComputeStereoMatches
(reference src/Frame.cc:1102-1358) all trigger.
"""
import numpy as np

SEED0 = 0x0BB5


def _scene(rng, w, h, nrect):
    x = rng.integers(-20, w, nrect)
    y = rng.integers(-20, h, nrect)
    rw = rng.integers(4, 41, nrect)
    rh = rng.integers(4, 41, nrect)
    contrast = rng.integers(20, 121, nrect) * rng.choice([-1, 1], nrect)
    disp = rng.integers(2, 61, nrect)
    return x, y, rw, rh, contrast, disp


def _render(w, h, scene, shift_sign, base_shift, contrast_div):
    x, y, rw, rh, contrast, disp = scene
    yy, xx = np.mgrid[0:h, 0:w]
    img = 110.0 + 50.0 * np.sin((xx + shift_sign * base_shift) / 97.0) * np.cos(yy / 71.0)
    for i in range(len(x)):
        x0 = int(x[i] - shift_sign * disp[i]); y0 = int(y[i])
        xa, xb = max(x0, 0), min(x0 + int(rw[i]), w)
        ya, yb = max(y0, 0), min(y0 + int(rh[i]), h)
        if xa < xb and ya < yb:
            img[ya:yb, xa:xb] += contrast[i] / contrast_div
    return img


def corner_field(w=752, h=480, seed=0, nrect=3000, noise=3.0, contrast_div=1.0):
    """S1 (contrast_div=1) / S2 low texture (contrast_div=6) single image, uint8 HxW."""
    rng = np.random.default_rng(SEED0 + seed)
    sc = _scene(rng, w, h, nrect)
    img = _render(w, h, sc, 0, 0, contrast_div) + rng.normal(0, noise, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def stereo_pair(w=752, h=480, seed=0, nrect=3000, noise=2.0, band=48, max_disp=60):
    """Rectified stereo pair (left, right).  One textured scene is rendered wider than the image; the left view
    is a window of it and the right view is the same window shifted by a per-band disparity (horizontal bands
    of `band` rows = depth layers, disparity 2..max_disp px) plus independent sensor noise.  Most keypoints
    therefore have a true match on the same row with a near-identical descriptor, band seams give occlusions,
    and the noise makes the SAD parabola fit non-trivial."""
    rng = np.random.default_rng(SEED0 + seed)
    pad = max_disp + 4
    sc = _scene(rng, w + pad, h, int(nrect * (w + pad) / w))
    scene = _render(w + pad, h, sc, 0, 0, 1.0)
    nb = (h + band - 1) // band
    disp = rng.integers(2, max_disp + 1, nb)
    left = scene[:, :w].copy()
    right = np.empty_like(left)
    for k in range(nb):
        y0, y1 = k * band, min((k + 1) * band, h)
        d = int(disp[k])
        right[y0:y1] = scene[y0:y1, d:d + w]
    left = left + rng.normal(0, noise, (h, w))
    right = right + rng.normal(0, noise, (h, w))
    cv = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return cv(left), cv(right)


def sparse_corners(w=752, h=480, seed=0, ncorner=40):
    """S3: flat image with a few isolated corners (empty cells, N < nfeatures)."""
    rng = np.random.default_rng(SEED0 + 7919 + seed)
    img = np.full((h, w), 100, np.uint8)
    for _ in range(ncorner):
        x0 = int(rng.integers(25, w - 60)); y0 = int(rng.integers(25, h - 60))
        img[y0:y0 + int(rng.integers(8, 30)), x0:x0 + int(rng.integers(8, 30))] = int(rng.integers(150, 255))
    return img


def uniform_noise(w=752, h=480, seed=0):
    """S4: uniform random bytes (stress: a candidate at almost every local maximum)."""
    rng = np.random.default_rng(SEED0 + 104729 + seed)
    return rng.integers(0, 256, (h, w), dtype=np.uint8)


def pink_noise(w=752, h=480, seed=0, beta=1.6, contrast=60.0):
    """Natural-image-like texture: Gaussian noise shaped to a 1/f^beta amplitude spectrum (smooth gradients, corners at every scale,
    a FAST corner rate of a few percent instead of the >20 % of corner_field), 8-bit."""
    rng = np.random.default_rng(0x91E5 + seed)
    fy = np.fft.fftfreq(h)[:, None]; fx = np.fft.rfftfreq(w)[None, :]
    f = np.sqrt(fx * fx + fy * fy); f[0, 0] = 1.0
    spec = (rng.standard_normal((h, w // 2 + 1)) + 1j * rng.standard_normal((h, w // 2 + 1))) / f ** beta
    spec[0, 0] = 0.0
    img = np.fft.irfft2(spec, s=(h, w))
    img = (img - img.mean()) / (img.std() + 1e-12)
    return np.clip(128.0 + contrast * img, 0, 255).astype(np.uint8)


def periodic_stereo_pair(w=752, h=480, seed=0, period=(48, 240), disparity=17, nrect=40):
    """Adversarial input for the stereo row search: a noise-free texture that repeats every `period[0]` columns (and `period[1]` rows),
    right view = left view shifted by `disparity`.  Every keypoint of the right image then has exact copies - identical descriptors - on the
    same row at every multiple of the period, so a left keypoint sees several candidates with the same minimal Hamming distance and the
    result depends on the tie rule alone (reference src/Frame.cc:1195-1226: the lowest right index wins)."""
    rng = np.random.default_rng(0x7E5 + seed)
    px, py = period
    tile = np.full((py, px), 110.0)
    for _ in range(nrect):
        x0, y0 = int(rng.integers(0, px)), int(rng.integers(0, py))
        rw, rh = int(rng.integers(4, max(px // 2, 6))), int(rng.integers(4, 30))
        c = float(rng.integers(25, 110)) * (1 if rng.random() < 0.5 else -1)
        ys = (np.arange(y0, y0 + rh) % py)[:, None]; xs = (np.arange(x0, x0 + rw) % px)[None, :]
        tile[ys, xs] += c
    reps = ((h + py - 1) // py + 1, (w + disparity + px - 1) // px + 1)
    scene = np.tile(tile, reps)
    cv = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return cv(scene[:h, :w]), cv(scene[:h, disparity:disparity + w])


def _natural_scene(rng, w, h, beta, contrast, nedge, noise):
    """float scene: 1/f^beta Gaussian field + a sparse set of occluding polygons (step edges with corners, like furniture / window frames)
    + sensor noise"""
    fy = np.fft.fftfreq(h)[:, None]; fx = np.fft.rfftfreq(w)[None, :]
    f = np.sqrt(fx * fx + fy * fy); f[0, 0] = 1.0
    spec = (rng.standard_normal((h, w // 2 + 1)) + 1j * rng.standard_normal((h, w // 2 + 1))) / f ** beta
    spec[0, 0] = 0.0
    img = np.fft.irfft2(spec, s=(h, w))
    img = 120.0 + contrast * (img - img.mean()) / (img.std() + 1e-12)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(nedge):
        # a rotated rectangle with its own brightness offset and a little texture of its own
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        a, b = rng.uniform(10, 90), rng.uniform(10, 90)
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        m = (np.abs(u) < a) & (np.abs(v) < b)
        img[m] += rng.choice([-1.0, 1.0]) * rng.uniform(12, 70)
    return img + rng.normal(0, noise, (h, w))


def natural(w=752, h=480, seed=0, beta=2.0, contrast=38.0, nedge=100, noise=1.5):
    """Camera-like image: 1/f^2 amplitude spectrum (power spectrum 1/f^4 is too smooth; natural images sit near amplitude 1/f - 1/f^2 over the
    band FAST looks at), sparse occluding rectangles at random orientations, mild sensor noise.  1-5 % of the pixels are FAST-9/16 corners at
    t = 7 (synth.fast_corner_density), an order of magnitude fewer than corner_field - the density regime of EuRoC-like imagery."""
    rng = np.random.default_rng(0x4A70 + seed)
    return np.clip(np.rint(_natural_scene(rng, w, h, beta, contrast, int(nedge * w * h / (752 * 480)), noise)), 0, 255).astype(np.uint8)


def natural_stereo_pair(w=752, h=480, seed=0, band=48, max_disp=60, **kw):
    """Rectified pair with natural() statistics: one wider scene, the right view shifted by a per-band disparity, independent noise per view."""
    rng = np.random.default_rng(0x4A71 + seed)
    pad = max_disp + 4
    noise = kw.pop("noise", 1.5)
    scene = _natural_scene(rng, w + pad, h, kw.pop("beta", 2.0), kw.pop("contrast", 38.0), int(kw.pop("nedge", 100) * (w + pad) * h / (752 * 480)), 0.0)
    nb = (h + band - 1) // band
    disp = rng.integers(2, max_disp + 1, nb)
    left = scene[:, :w].copy(); right = np.empty_like(left)
    for k in range(nb):
        y0, y1 = k * band, min((k + 1) * band, h)
        d = int(disp[k])
        right[y0:y1] = scene[y0:y1, d:d + w]
    cv = lambda a: np.clip(np.rint(a + rng.normal(0, noise, a.shape)), 0, 255).astype(np.uint8)
    return cv(left), cv(right)


def threshold_blocks(w=376, h=240, seed=0, block=37):
    """Dots on a flat background, one kind per block of about one FAST cell: nothing; weak dots (contrast 9-17: corners at minThFAST = 7 only -
    the cell's second FAST run, src/ORBextractor.cc:1143-1148, decides); strong dots (only the first run counts); strong dot PAIRS (two
    adjacent pixels with equal scores: neither is a strict 3x3 maximum, so a cell can hold corners at iniThFAST and still come out empty
    and fall back); and mixtures of these."""
    rng = np.random.default_rng(seed)
    a = np.full((h, w), 60, np.uint8)
    for by in range(0, h - block + 1, block):
        for bx in range(0, w - block + 1, block):
            kind = int(rng.integers(0, 6))
            def spot():
                return by + int(rng.integers(4, block - 5)), bx + int(rng.integers(4, block - 5))
            if kind in (1, 3, 4):                       # weak single dots
                for _ in range(int(rng.integers(1, 4))):
                    y, x = spot(); a[y, x] = 60 + int(rng.integers(9, 18))
            if kind in (2, 3):                          # strong single dots
                for _ in range(int(rng.integers(1, 3))):
                    y, x = spot(); a[y, x] = 60 + int(rng.integers(40, 150))
            if kind in (4, 5):                          # strong tied pairs
                y, x = spot(); v = 60 + int(rng.integers(40, 150)); a[y, x] = v; a[y, x + 1] = v
    return a


_RING = ((0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3))


def fast_corner_density(img, t=7):
    """Fraction of the pixels (3-px border excluded) that are FAST-9/16 corners at threshold t, before non-maximum suppression: the workload
    statistic the FAST kernel's cost depends on (numpy, the segment test as defined; not used by any product path)."""
    a = img.astype(np.int16)
    h, w = a.shape
    c = a[3:h - 3, 3:w - 3]
    br = np.stack([a[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] > c + t for dx, dy in _RING])
    dk = np.stack([a[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] < c - t for dx, dy in _RING])

    def arc9(m):
        m2 = np.concatenate([m, m[:8]])
        run = np.ones_like(m[0])
        acc = np.zeros_like(m[0])
        for s in range(16):
            run = m2[s].copy()
            for k in range(1, 9):
                run &= m2[s + k]
            acc |= run
        return acc
    return float((arc9(br) | arc9(dk)).mean())
