"""ctypes mirrors of the read-only structure-of-arrays views in include/orbx.h (OrbmFrameView, OrbmMapPointView,
OrbmLastFrameView, OrbmKeyFrameView, OrbmProjectedPointView) and helpers that build them from numpy arrays."""
import ctypes as C

import numpy as np

from ._lib import KP_DTYPE

_vp, _i, _f = C.c_void_p, C.c_int, C.c_float


class FrameView(C.Structure):
    _fields_ = [("N", _i), ("keys_un", _vp), ("desc", _vp), ("u_right", _vp), ("occupied", _vp),
                ("min_x", _f), ("min_y", _f), ("max_x", _f), ("max_y", _f), ("grid_w_inv", _f), ("grid_h_inv", _f),
                ("mbf", _f), ("nlevels", _i), ("scale_factors", _vp)]


class MapPointView(C.Structure):
    _fields_ = [("M", _i), ("in_view", _vp), ("proj_x", _vp), ("proj_y", _vp), ("proj_xr", _vp), ("scale_level", _vp),
                ("view_cos", _vp), ("track_depth", _vp), ("is_bad", _vp), ("has_obs", _vp), ("desc", _vp)]


class LastFrameView(C.Structure):
    _fields_ = [("N", _i), ("valid", _vp), ("proj_u", _vp), ("proj_v", _vp), ("inv_z", _vp), ("octave", _vp),
                ("angle", _vp), ("has_obs", _vp), ("desc", _vp)]


class KeyFrameView(C.Structure):
    _fields_ = [("N", _i), ("keys_un", _vp), ("desc", _vp), ("u_right", _vp), ("has_map_point", _vp), ("fv_nodes", _i),
                ("fv_node_id", _vp), ("fv_start", _vp), ("fv_feat", _vp), ("nlevels", _i), ("scale_factors", _vp),
                ("level_sigma2", _vp)]


class FisheyeFrameView(C.Structure):
    _fields_ = [("left", FrameView), ("right", FrameView), ("left_to_right", _vp), ("right_to_left", _vp)]


class MapPointRightView(C.Structure):
    _fields_ = [("in_view_r", _vp), ("proj_xr", _vp), ("proj_yr", _vp), ("scale_level_r", _vp), ("view_cos_r", _vp)]


class ProjectedPointView(C.Structure):
    _fields_ = [("M", _i), ("valid", _vp), ("u", _vp), ("v", _vp), ("ur", _vp), ("pred_level", _vp), ("angle", _vp), ("desc", _vp)]


def _arr(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype)


def _ptr(a):
    return None if a is None else a.ctypes.data


class Held:
    """A ctypes view plus the numpy arrays that back its pointers."""

    def __init__(self, view, keep):
        self.view, self.keep = view, keep

    def ref(self):
        return C.byref(self.view)


def frame_view(keys_un, desc, scale_factors, width, height, u_right=None, occupied=None, mbf=0.0, bounds=None):
    """bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY); default: the image rectangle (no distortion, src/Frame.cc:1083-1090).
    Grid cell inverses as in src/Frame.cc:187-189: FRAME_GRID_COLS / (mnMaxX - mnMinX)."""
    k = _arr(keys_un, KP_DTYPE); d = _arr(desc, np.uint8); s = _arr(scale_factors, np.float32)
    u = _arr(u_right, np.float32); o = _arr(occupied, np.uint8)
    mnx, mxx, mny, mxy = bounds if bounds else (0.0, float(width), 0.0, float(height))
    gw = np.float32(64.0) / np.float32(np.float32(mxx) - np.float32(mnx)); gh = np.float32(48.0) / np.float32(np.float32(mxy) - np.float32(mny))
    v = FrameView(len(k), _ptr(k), _ptr(d), _ptr(u), _ptr(o), mnx, mny, mxx, mxy, float(gw), float(gh), float(mbf), len(s), _ptr(s))
    return Held(v, (k, d, s, u, o))


def map_point_view(in_view, proj_x, proj_y, proj_xr, scale_level, view_cos, track_depth, is_bad, has_obs, desc):
    a = [_arr(in_view, np.uint8), _arr(proj_x, np.float32), _arr(proj_y, np.float32), _arr(proj_xr, np.float32),
         _arr(scale_level, np.int32), _arr(view_cos, np.float32), _arr(track_depth, np.float32), _arr(is_bad, np.uint8),
         _arr(has_obs, np.uint8), _arr(desc, np.uint8)]
    return Held(MapPointView(len(a[0]), *[_ptr(x) for x in a]), a)


def last_frame_view(valid, proj_u, proj_v, inv_z, octave, angle, has_obs, desc):
    a = [_arr(valid, np.uint8), _arr(proj_u, np.float32), _arr(proj_v, np.float32), _arr(inv_z, np.float32),
         _arr(octave, np.int32), _arr(angle, np.float32), _arr(has_obs, np.uint8), _arr(desc, np.uint8)]
    return Held(LastFrameView(len(a[0]), *[_ptr(x) for x in a]), a)


def key_frame_view(keys_un, desc, scale_factors, level_sigma2, fv_node_id, fv_start, fv_feat, u_right=None, has_map_point=None):
    k = _arr(keys_un, KP_DTYPE); d = _arr(desc, np.uint8); s = _arr(scale_factors, np.float32); g = _arr(level_sigma2, np.float32)
    ni = _arr(fv_node_id, np.uint32); st = _arr(fv_start, np.int32); ft = _arr(fv_feat, np.uint32)
    u = _arr(u_right, np.float32); m = _arr(has_map_point, np.uint8)
    v = KeyFrameView(len(k), _ptr(k), _ptr(d), _ptr(u), _ptr(m), len(ni), _ptr(ni), _ptr(st), _ptr(ft), len(s), _ptr(s), _ptr(g))
    return Held(v, (k, d, s, g, ni, st, ft, u, m))


def projected_point_view(valid, u, v, pred_level, desc, ur=None, angle=None):
    """OrbmProjectedPointView: map points whose geometry (projection, gates, PredictScale) the caller has already evaluated."""
    a = [_arr(valid, np.uint8), _arr(u, np.float32), _arr(v, np.float32), _arr(ur, np.float32), _arr(pred_level, np.int32),
         _arr(angle, np.float32), _arr(desc, np.uint8)]
    return Held(ProjectedPointView(len(a[0]), *[_ptr(x) for x in a]), a)


def fisheye_frame_view(left, right, left_to_right=None, right_to_left=None):
    """OrbmFisheyeFrameView from two frame_view() results (camera 1 / camera 2 of a Frame with Nleft != -1)."""
    l2r = _arr(left_to_right, np.int32); r2l = _arr(right_to_left, np.int32)
    v = FisheyeFrameView(left.view, right.view, _ptr(l2r), _ptr(r2l))
    return Held(v, (left, right, l2r, r2l))


def map_point_right_view(in_view_r, proj_xr, proj_yr, scale_level_r, view_cos_r):
    a = [_arr(in_view_r, np.uint8), _arr(proj_xr, np.float32), _arr(proj_yr, np.float32), _arr(scale_level_r, np.int32), _arr(view_cos_r, np.float32)]
    return Held(MapPointRightView(*[_ptr(x) for x in a]), a)
