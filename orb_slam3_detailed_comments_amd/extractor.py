"""Host-side mirror of ``ORB_SLAM3::ORBextractor`` (reference include/ORBextractor.h:43-109) on top of the
HIP library: same constructor arguments, same call convention and return value (monoIndex, -1 on an empty
image), same getters; plus the batched form that the MI355X design is built around."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE


class InputSpec(C.Structure):
    """OrbxInputSpec (include/orbx.h)"""
    _fields_ = [("channels", C.c_int), ("rgb", C.c_int), ("gray_variant", C.c_int),
                ("geometry", C.c_int), ("out_w", C.c_int), ("out_h", C.c_int),
                ("map_x", C.c_void_p), ("map_y", C.c_void_p)]


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1   # include/ORBextractor.h:47 (only FAST_SCORE is ever computed)

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device_id=0, lib=None):
        self._lib = lib or _lib.load_hip()
        self._h = C.c_void_p()
        self._lib.check(self._lib.L.orbx_create(C.byref(self._h), int(nfeatures), float(scaleFactor), int(nlevels),
                                               int(iniThFAST), int(minThFAST), int(device_id)))
        self.nfeatures, self.nlevels = int(nfeatures), int(nlevels)
        self._shape = None
        self._B = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._fetch_buf = None
            for p in getattr(self, "_pinned", []):          # page-locked buffers handed out by pinned_empty()
                self._lib.L.orbx_host_free(self._h, p)
            self._pinned = []
            self._lib.L.orbx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (include/ORBextractor.h:61-81) ----
    def GetLevels(self):
        return self._lib.L.orbx_get_levels(self._h)

    def GetScaleFactor(self):
        return self._lib.L.orbx_get_scale_factor(self._h)

    def _tables(self):
        nl = self.nlevels
        f = [np.zeros(nl, np.float32) for _ in range(4)]
        q = np.zeros(nl, np.int32); um = np.zeros(16, np.int32)
        self._lib.check(self._lib.L.orbx_get_level_tables(self._h, *[a.ctypes.data for a in f], q.ctypes.data, um.ctypes.data))
        return f, q, um

    def GetScaleFactors(self):
        return self._tables()[0][0]

    def GetInverseScaleFactors(self):
        return self._tables()[0][1]

    def GetScaleSigmaSquares(self):
        return self._tables()[0][2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[0][3]

    def features_per_level(self):
        return self._tables()[1]

    def umax(self):
        return self._tables()[2]

    def set_gaussian_taps(self, variant):
        self._lib.check(self._lib.L.orbx_set_gaussian_taps(self._h, int(variant)))

    def max_keypoints(self):
        return self._lib.L.orbx_max_keypoints(self._h)

    def set_input(self, channels=1, rgb=True, gray_variant=0, remap=None, resize=None):
        """The reference's steps between the camera driver and the extractor (orbx_set_input): `remap=(map_x, map_y)` = stereo
        rectification cv::remap(..., INTER_LINEAR) with CV_32FC1 maps (src/System.cc:286-293); `resize=(new_w, new_h)` = cv::resize to
        Settings::newImSize (:295-297); channels 3/4 + rgb = the cvtColor of Tracking::GrabImage* (src/Tracking.cc:1532-1560).
        Frames are then passed as [B,H,W] (1 channel) or [B,H,W,C].  set_input(None) restores plain 8UC1 input."""
        import ctypes as C
        Spec = InputSpec
        if channels is None:
            self._lib.check(self._lib.L.orbx_set_input(self._h, None)); self._in_channels = 1
            return
        sp = Spec(int(channels), int(bool(rgb)), int(gray_variant), 0, 0, 0, None, None)
        keep = None
        if remap is not None:
            mx = np.ascontiguousarray(remap[0], np.float32); my = np.ascontiguousarray(remap[1], np.float32)
            assert mx.shape == my.shape and mx.ndim == 2
            sp.geometry, sp.out_w, sp.out_h, sp.map_x, sp.map_y = 1, mx.shape[1], mx.shape[0], mx.ctypes.data, my.ctypes.data
            keep = (mx, my)
        elif resize is not None:
            sp.geometry, sp.out_w, sp.out_h = 2, int(resize[0]), int(resize[1])
        self._lib.check(self._lib.L.orbx_set_input(self._h, C.byref(sp)))
        del keep
        self._in_channels = int(channels)

    # ---- ORBextractor::operator() (src/ORBextractor.cc:1557) ----
    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """Returns (monoIndex, keypoints[N] structured array, descriptors[N,32] uint8); monoIndex == -1 and empty
        outputs for an empty image, like the reference.  `mask` is ignored, like the reference."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (src/ORBextractor.cc:1567)"
        res = self.extract_batch(image[None], vLappingArea)
        return res[0]

    def enqueue(self, images, lap=(0, 0), device_ptr=None, shape=None, stride=None, image_stride=None):
        """Asynchronous batched extraction.  images: uint8 [B,H,W] host array, or (device_ptr, shape=(B,H,W))."""
        if device_ptr is None:
            images = np.ascontiguousarray(images, np.uint8)
            B, H, W = images.shape[:3]
            assert images.ndim == 3 or images.shape[3] == getattr(self, "_in_channels", 1), "channel count differs from set_input()"
            ptr, st, ist, ondev = images.ctypes.data, images.strides[1], images.strides[0], 0
            self._keep = images
        else:
            B, H, W = shape
            ptr, st, ist, ondev = device_ptr, stride or W, image_stride or (stride or W) * H, 1
        self._lib.check(self._lib.L.orbx_extract_batch(self._h, B, ptr, W, H, st, ist, ondev, int(lap[0]), int(lap[1])))
        self._shape, self._B = (H, W), B
        return B

    def fetch(self):
        B, cap = self._B, self.max_keypoints()
        buf = getattr(self, "_fetch_buf", None)
        if buf is None or buf[0] != (B, cap):      # page-locked staging, reused across calls (the device layout is [B, cap])
            if buf is not None:                    # a new batch shape replaces the staging pair: give the old pages back
                self._fetch_buf = None
                for a in buf[1:]:
                    self.pinned_free(a)
            buf = ((B, cap), self.pinned_empty((B, cap), KP_DTYPE), self.pinned_empty((B, cap, 32), np.uint8))
            self._fetch_buf = buf
        kps, desc = buf[1], buf[2]
        n = np.zeros(B, np.int32); mono = np.zeros(B, np.int32)
        self._lib.check(self._lib.L.orbx_fetch(self._h, kps.ctypes.data, desc.ctypes.data, cap, n.ctypes.data, mono.ctypes.data))
        return [(int(mono[b]), kps[b, :n[b]].copy(), desc[b, :n[b]].copy()) for b in range(B)]

    def extract_batch(self, images, lap=(0, 0)):
        self.enqueue(images, lap)
        return self.fetch()

    def sync(self):
        self._lib.check(self._lib.L.orbx_sync(self._h))

    # ---- mvImagePyramid (include/ORBextractor.h:83) ----
    def pyramid_level(self, level, image_index=0, blurred=False):
        w, h = C.c_int(), C.c_int()
        self._lib.check(self._lib.L.orbx_pyramid_level(self._h, image_index, level, int(blurred), None, 0, C.byref(w), C.byref(h)))
        a = np.zeros((h.value, w.value), np.uint8)
        self._lib.check(self._lib.L.orbx_pyramid_level(self._h, image_index, level, int(blurred), a.ctypes.data, w.value, C.byref(w), C.byref(h)))
        return a

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    # ---- stage probes / profiling ----
    def debug_candidates(self, level, image_index=0):
        cap = 1 << 18
        a = np.zeros((cap, 3), np.int32)
        n = self._lib.L.orbx_debug_candidates(self._h, image_index, level, a.ctypes.data, cap)
        if n > cap:                                   # the call reports the full count and fills what fits: ask again with room for all of it
            a = np.zeros((n, 3), np.int32)
            n = self._lib.L.orbx_debug_candidates(self._h, image_index, level, a.ctypes.data, n)
        if n < 0:
            self._lib.check(n)
        return a[:n].copy()

    def debug_level_keys(self, level, image_index=0):
        cap = self.max_keypoints()
        a = np.zeros((cap, 3), np.int32)
        n = self._lib.L.orbx_debug_level_keys(self._h, image_index, level, a.ctypes.data, cap)
        if n < 0:
            self._lib.check(n)
        return a[:n].copy()

    def debug_stereo_flags(self, flags):
        """Test switches of the stereo row search (orbx_debug_stereo_flags)."""
        self._lib.check(self._lib.L.orbx_debug_stereo_flags(self._h, int(flags)))

    def pyramid_mode(self, mode):
        """0 = by batch size (default), 1 = one launch per pyramid level, 2 = all levels in one launch (orbx_set_pyramid_mode)."""
        self._lib.check(self._lib.L.orbx_set_pyramid_mode(self._h, int(mode)))

    def set_small_batch_forms(self, on):
        """True (default): batches of up to 32 images run the blur strips and the FAST cells in one launch on one stream; False: the large-batch
        form (two launches on two streams) at every batch size (orbx_set_small_batch_forms).  Bit-identical outputs."""
        self._lib.check(self._lib.L.orbx_set_small_batch_forms(self._h, 1 if on else 0))

    def debug_quadtree_lds_nodes(self, max_nodes):
        """Test hook: levels whose quadtree may hold more than max_nodes nodes keep their node lists in the global node pool instead of LDS
        (orbx_debug_quadtree_lds_nodes; 0 = every level).  Bit-identical outputs."""
        self._lib.check(self._lib.L.orbx_debug_quadtree_lds_nodes(self._h, int(max_nodes)))

    def debug_quadtree_pool_levels(self):
        """Number of pyramid levels whose quadtree the last extraction ran in the pool form (orbx_debug_quadtree_pool_levels)."""
        return int(self._lib.L.orbx_debug_quadtree_pool_levels(self._h))

    def graph_replay(self, on=True):
        """Replay the extraction pipeline as one hipGraph (small-batch latency)."""
        self._lib.check(self._lib.L.orbx_set_graph_replay(self._h, int(on)))

    def profile(self, on=True, serial=False):
        self._lib.check(self._lib.L.orbx_profile_enable(self._h, 2 if (on and serial) else int(on)))

    def stage_ms(self):
        ms = np.zeros(_lib.NSTAGES, np.float32)
        self._lib.check(self._lib.L.orbx_profile_get(self._h, ms.ctypes.data))
        return dict(zip(self._lib.stage_names(), ms.tolist()))

    # ---- device memory helpers ----
    def device_upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._lib.check(self._lib.L.orbx_device_alloc(self._h, arr.nbytes, C.byref(p)))
        self._lib.check(self._lib.L.orbx_device_upload(self._h, p, arr.ctypes.data, arr.nbytes))
        return p

    def set_undistort(self, K=None, dist=None, opencv_variant=0):
        """Frame::UndistortKeyPoints on the device (orbx_set_undistort): K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2[, k3]); None switches it off."""
        if K is None or dist is None:
            self._lib.check(self._lib.L.orbx_set_undistort(self._h, None, None, 0, 0)); return
        k = np.ascontiguousarray(K, np.float32); d = np.ascontiguousarray(dist, np.float32)
        self._lib.check(self._lib.L.orbx_set_undistort(self._h, k.ctypes.data, d.ctypes.data, len(d), int(opencv_variant)))

    def fetch_undistorted(self):
        """mvKeysUn of the last batch: [B, cap] keypoint records (rows beyond the count of a frame are unspecified)"""
        cap = self.max_keypoints()
        a = np.zeros((self._B, cap), KP_DTYPE)
        self._lib.check(self._lib.L.orbx_fetch_undistorted(self._h, a.ctypes.data, cap))
        return a

    def undistorted_bounds(self, width, height):
        """Frame::ComputeImageBounds: (mnMinX, mnMaxX, mnMinY, mnMaxY)"""
        o = np.zeros(4, np.float32)
        self._lib.check(self._lib.L.orbx_undistorted_bounds(self._h, int(width), int(height), o.ctypes.data))
        return tuple(float(v) for v in o)

    def input_upload(self, images):
        """Zero-copy input: writes the host batch [B, H, W] straight into pyramid level 0 of this handle (orbx_input_buffer / orbx_input_upload) and
        returns (device_ptr, shape, stride, image_stride) for enqueue(device_ptr=...): the extraction then reads level 0 in place, without the
        import pass.  The frames stay resident (level 0 is never written by the extraction)."""
        images = np.ascontiguousarray(images, np.uint8)
        B, H, W = images.shape
        self._lib.check(self._lib.L.orbx_input_upload(self._h, B, images.ctypes.data, W, H, images.strides[1], images.strides[0]))
        p = C.c_void_p(); st = C.c_int(); ist = C.c_size_t()
        self._lib.check(self._lib.L.orbx_input_buffer(self._h, W, H, B, C.byref(p), C.byref(st), C.byref(ist)))
        return p, (B, H, W), st.value, ist.value

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._lib.check(self._lib.L.orbx_device_alloc(self._h, int(nbytes), C.byref(p)))
        return p

    def device_upload_async(self, dptr, arr):
        """Upload `arr` (page-locked: pinned_empty) into the device buffer on the handle's copy stream; the next enqueue(device_ptr=...)
        waits for it on the device."""
        self._lib.check(self._lib.L.orbx_device_upload_async(self._h, dptr, arr.ctypes.data, arr.nbytes))

    def pinned_empty(self, shape, dtype):
        """numpy array backed by page-locked host memory (fast D2H target for fetch())."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._lib.check(self._lib.L.orbx_host_alloc(self._h, max(nbytes, 1), C.byref(p)))
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", []) + [p]
        return arr

    def pinned_free(self, arr):
        """Release a pinned_empty() array (the caller must not touch `arr` afterwards); close() releases whatever is left."""
        addr = arr.ctypes.data
        for p in getattr(self, "_pinned", []):
            if p.value == addr:
                self._pinned.remove(p)
                self._lib.L.orbx_host_free(self._h, p)
                return

    def device_free(self, p):
        self._lib.L.orbx_device_free(self._h, p)
