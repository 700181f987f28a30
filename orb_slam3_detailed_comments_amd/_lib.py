"""ctypes binding of the C ABI in include/orbx.h.

The product library is the hipcc-built, in-tree ``liborbx_hip.so``.  There is NO CPU fallback: if the
library is missing or no GPU is usable, loading / creating an extractor raises.  (Tests may bind the same
class to ``tests/emu/liborbx_emu.so`` — the kernel sources compiled for a CPU SIMT emulator — by passing an
explicit path; the package itself never does.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "liborbx_hip.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28
NSTAGES = 8

ORBX_OK, ORBX_E_EMPTY, ORBX_E_ARG, ORBX_E_DEVICE, ORBX_E_CAPACITY, ORBX_E_INTERNAL = 0, -1, -2, -3, -4, -5

# every symbol include/orbx.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "orbx_device_count", "orbx_create", "orbx_destroy", "orbx_set_gaussian_taps", "orbx_reserve",
    "orbx_get_levels", "orbx_get_scale_factor", "orbx_get_level_tables", "orbx_max_keypoints",
    "orbx_extract", "orbx_extract_batch", "orbx_set_input", "orbx_fetch", "orbx_sync", "orbx_pyramid_level", "orbx_pyramid_fetch",
    "orbx_device_alloc", "orbx_device_free", "orbx_device_upload", "orbx_device_upload_async", "orbx_set_undistort", "orbx_fetch_undistorted", "orbx_undistorted_bounds", "orbx_input_buffer", "orbx_input_upload", "orbx_device_outputs", "orbx_device_snapshot", "orbx_device_id", "orbx_host_alloc", "orbx_host_free", "orbx_set_graph_replay", "orbx_set_pyramid_mode", "orbx_set_small_batch_forms", "orbx_profile_enable",
    "orbx_profile_get", "orbx_stage_name", "orbx_debug_candidates", "orbx_debug_level_keys", "orbx_debug_quadtree_profile", "orbx_set_host_wait", "orbx_debug_quadtree_lds_nodes", "orbx_debug_quadtree_pool_levels", "orbx_debug_simd_selftest", "orbx_debug_stereo_flags", "orbx_debug_live_resources",
    "orbm_hamming_matrix", "orbm_stereo_match", "orbm_stereo_fetch", "orbm_knn2", "orbm_knn2_fetch", "orbm_stereo_fisheye", "orbm_stereo_fisheye_fetch", "orbm_search_for_triangulation_kb8", "orbm_is_in_frustum", "orbm_is_in_frustum_rig", "orbm_search_local_points_fisheye", "orbm_search_local_points",
    "orbm_get_features_in_area", "orbm_search_by_projection_mappoints", "orbm_search_by_projection_frame",
    "orbm_search_for_triangulation", "orbm_search_by_bow", "orbm_search_by_bow_batch", "orbm_keyframe_create", "orbm_points_create", "orbm_points_destroy", "orbm_search_local_points_resident", "orbm_stereo_from_depth", "orbm_search_local_points_batch", "orbm_search_local_points_fetch", "orbm_search_by_projection_lastframe_batch", "orbm_search_by_projection_keyframe_batch", "orbm_keyframe_destroy", "orbm_search_for_triangulation_resident", "orbm_search_for_triangulation_resident_kb8", "orbm_search_by_bow_resident", "orbm_search_by_bow_fisheye", "orbm_search_for_initialization", "orbm_area_search_batch",
    "orbm_project_points", "orbm_search_by_projection_sim3", "orbm_search_by_projection_keyframe", "orbm_fuse_candidates", "orbm_search_by_sim3", "orbm_distinctive_descriptors",
    "orbm_search_by_projection_mappoints_fisheye", "orbm_search_by_projection_frame_fisheye", "orbm_search_for_triangulation_batch",
    "orbv_create", "orbv_load_text", "orbv_destroy", "orbv_words", "orbv_transform", "orbv_transform_extracted", "orbv_fetch", "orbm_search_by_bow_frames_batch",
    "orbx_comm_unique_id", "orbx_comm_create", "orbx_comm_adopt", "orbx_comm_destroy", "orbx_comm_world", "orbx_comm_rank", "orbx_allgather_descriptors", "orbx_comm_wait",
    "orbx_comm_fetch",
    "orbx_last_error",
]


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbx error %d: %s" % (code, msg))
        self.code = code


class OrbxLib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise RuntimeError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
        self.path = path
        L = self.L = C.CDLL(path)
        vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        ip = C.POINTER(C.c_int)
        L.orbx_device_count.restype = i
        L.orbx_create.argtypes = [C.POINTER(vp), i, f, i, i, i, i]
        L.orbx_destroy.argtypes = [vp]; L.orbx_destroy.restype = None
        L.orbx_set_gaussian_taps.argtypes = [vp, i]
        L.orbx_reserve.argtypes = [vp, i, i, i]
        L.orbx_get_levels.argtypes = [vp]
        L.orbx_get_scale_factor.argtypes = [vp]; L.orbx_get_scale_factor.restype = f
        L.orbx_get_level_tables.argtypes = [vp] + [vp] * 6
        L.orbx_max_keypoints.argtypes = [vp]
        L.orbx_extract.argtypes = [vp, vp, i, i, i, i, i, vp, vp, i, ip, ip]
        L.orbx_extract_batch.argtypes = [vp, i, vp, i, i, i, sz, i, i, i]
        L.orbx_set_input.argtypes = [vp, vp]
        L.orbx_fetch.argtypes = [vp, vp, vp, i, vp, vp]
        L.orbx_sync.argtypes = [vp]
        L.orbx_pyramid_level.argtypes = [vp, i, i, i, vp, i, ip, ip]
        L.orbx_pyramid_fetch.argtypes = [vp, i, i, vp, vp]
        L.orbx_device_alloc.argtypes = [vp, sz, C.POINTER(vp)]
        L.orbx_device_free.argtypes = [vp, vp]
        L.orbx_device_upload.argtypes = [vp, vp, vp, sz]
        L.orbx_device_upload_async.argtypes = [vp, vp, vp, sz]
        L.orbx_device_outputs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), ip, ip]
        L.orbx_set_undistort.argtypes = [vp, vp, vp, i, i]
        L.orbx_fetch_undistorted.argtypes = [vp, vp, i]
        L.orbx_undistorted_bounds.argtypes = [vp, i, i, vp]
        L.orbx_input_buffer.argtypes = [vp, i, i, i, C.POINTER(vp), ip, C.POINTER(sz)]
        L.orbx_input_upload.argtypes = [vp, i, vp, i, i, i, sz]
        L.orbx_device_snapshot.argtypes = [vp, vp, vp]
        L.orbx_device_id.argtypes = [vp]
        L.orbx_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
        L.orbx_host_free.argtypes = [vp, vp]
        L.orbx_set_graph_replay.argtypes = [vp, i]
        L.orbx_set_pyramid_mode.argtypes = [vp, i]
        L.orbx_set_small_batch_forms.argtypes = [vp, i]
        L.orbx_profile_enable.argtypes = [vp, i]
        L.orbx_profile_get.argtypes = [vp, vp]
        L.orbx_stage_name.argtypes = [i]; L.orbx_stage_name.restype = C.c_char_p
        L.orbx_debug_candidates.argtypes = [vp, i, i, vp, i]
        L.orbx_debug_level_keys.argtypes = [vp, i, i, vp, i]
        L.orbx_debug_quadtree_profile.argtypes = [vp, vp]
        L.orbx_debug_quadtree_lds_nodes.argtypes = [vp, i]
        L.orbx_set_host_wait.argtypes = [i, i]
        L.orbx_debug_quadtree_pool_levels.argtypes = [vp]
        L.orbx_debug_simd_selftest.argtypes = [vp, vp, vp, vp, i, vp]
        L.orbx_debug_stereo_flags.argtypes = [vp, i]
        L.orbx_debug_live_resources.argtypes = [vp]
        L.orbm_hamming_matrix.argtypes = [vp, vp, i, vp, i, vp]
        L.orbm_stereo_match.argtypes = [vp, i, vp, i, i, f, f]
        L.orbm_stereo_fetch.argtypes = [vp, i, vp, vp, i, vp]
        L.orbm_knn2.argtypes = [vp, i, vp, i, i]
        L.orbm_knn2_fetch.argtypes = [vp, i, vp, vp, vp, vp, vp, i]
        L.orbm_stereo_fisheye.argtypes = [vp, i, vp, i, i, vp]
        L.orbm_stereo_fisheye_fetch.argtypes = [vp, i, vp, vp, vp, vp, vp, i]
        L.orbm_search_for_triangulation_kb8.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp, ip]
        L.orbm_is_in_frustum.argtypes = [vp, vp, vp, f, vp]
        L.orbm_is_in_frustum_rig.argtypes = [vp, vp, vp, f, vp, vp]
        L.orbm_search_local_points_fisheye.argtypes = [vp, vp, vp, vp, f, f, i, f, f, vp, vp, vp, ip]
        L.orbm_search_local_points.argtypes = [vp, vp, vp, vp, f, f, i, f, f, vp, vp, ip]
        L.orbm_get_features_in_area.argtypes = [vp, vp, f, f, f, i, i, vp, i]
        L.orbm_search_by_projection_mappoints.argtypes = [vp, vp, vp, f, i, f, f, vp, ip]
        L.orbm_search_by_projection_frame.argtypes = [vp, vp, vp, f, i, i, i, vp, ip]
        L.orbm_search_for_triangulation.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp, ip]
        L.orbm_search_by_bow.argtypes = [vp, vp, vp, f, i, i, vp, ip]
        L.orbm_search_by_bow_fisheye.argtypes = [vp, vp, vp, i, f, i, vp, ip]
        L.orbm_search_by_bow_batch.argtypes = [vp, i, vp, vp, f, i, i, vp, vp]
        L.orbm_keyframe_create.argtypes = [vp, vp, vp]
        L.orbm_points_create.argtypes = [vp, vp, vp]
        L.orbm_points_destroy.argtypes = [vp]; L.orbm_points_destroy.restype = None
        L.orbm_search_local_points_resident.argtypes = [vp, vp, vp, vp, vp, vp, f, f, i, f, f, vp, vp, vp]
        L.orbm_stereo_from_depth.argtypes = [vp, i, i, vp, i, sz, i, f]
        L.orbm_search_local_points_batch.argtypes = [vp, i, i, vp, vp, vp, vp, vp, i, f, f, i, f, f, i]
        L.orbm_search_local_points_fetch.argtypes = [vp, vp, i, vp, vp]
        L.orbm_search_by_projection_lastframe_batch.argtypes = [vp, i, i, vp, vp, f, vp, vp, i, vp, i]
        L.orbm_search_by_projection_keyframe_batch.argtypes = [vp, i, i, vp, vp, f, i, i, vp]
        L.orbm_keyframe_destroy.argtypes = [vp]; L.orbm_keyframe_destroy.restype = None
        L.orbm_search_for_triangulation_resident.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, i, i, i, vp, vp]
        L.orbm_search_for_triangulation_resident_kb8.argtypes = [vp, vp, vp, i, vp, vp, vp, vp, i, i, i, vp, vp]
        L.orbm_search_by_bow_resident.argtypes = [vp, i, vp, vp, vp, vp, f, i, i, vp, vp]
        L.orbm_search_for_initialization.argtypes = [vp, vp, vp, vp, i, f, i, vp, ip]
        L.orbm_area_search_batch.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, vp, vp, i]
        L.orbm_project_points.argtypes = [vp, vp, vp, vp]
        L.orbm_search_by_projection_sim3.argtypes = [vp, vp, vp, f, f, vp, ip]
        L.orbm_search_by_projection_keyframe.argtypes = [vp, vp, vp, f, i, i, vp, ip]
        L.orbm_fuse_candidates.argtypes = [vp, vp, vp, f, i, vp, vp, vp]
        L.orbm_search_by_sim3.argtypes = [vp, vp, vp, vp, vp, f, vp, ip]
        L.orbm_search_by_projection_mappoints_fisheye.argtypes = [vp, vp, vp, vp, f, i, f, f, vp, ip]
        L.orbm_search_by_projection_frame_fisheye.argtypes = [vp, vp, vp, vp, vp, f, i, i, i, vp, ip]
        L.orbm_search_for_triangulation_batch.argtypes = [vp, vp, i, vp, vp, vp, i, i, i, vp, vp]
        L.orbm_distinctive_descriptors.argtypes = [vp, vp, vp, i, vp]
        L.orbv_create.argtypes = [vp, i, i, i, i, i, vp, vp, vp, vp, C.POINTER(vp)]
        L.orbv_load_text.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
        L.orbv_destroy.argtypes = [vp]; L.orbv_destroy.restype = None
        L.orbv_words.argtypes = [vp]
        L.orbv_transform.argtypes = [vp, vp, vp, i, i, vp, vp, vp, vp, ip, vp, vp, vp, ip]
        L.orbv_transform_extracted.argtypes = [vp, vp, i, i, i]
        L.orbv_fetch.argtypes = [vp, vp, i, vp, vp, i, vp, vp, ip, vp, vp, vp, ip]
        L.orbm_search_by_bow_frames_batch.argtypes = [vp, vp, i, i, vp, vp, f, i, vp, vp]
        L.orbx_comm_unique_id.argtypes = [vp]
        L.orbx_comm_create.argtypes = [C.POINTER(vp), i, i, vp, i]
        L.orbx_comm_adopt.argtypes = [C.POINTER(vp), vp, i, i, i]
        L.orbx_comm_destroy.argtypes = [vp]; L.orbx_comm_destroy.restype = None
        L.orbx_comm_world.argtypes = [vp]; L.orbx_comm_rank.argtypes = [vp]
        L.orbx_allgather_descriptors.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), ip, ip]
        L.orbx_comm_wait.argtypes = [vp]
        L.orbx_comm_fetch.argtypes = [vp, vp, vp]
        L.orbx_last_error.restype = C.c_char_p

    def check(self, rc):
        if rc != 0:
            raise OrbxError(rc, (self.L.orbx_last_error() or b"").decode())
        return rc

    def stage_names(self):
        return [self.L.orbx_stage_name(k).decode() for k in range(NSTAGES)]


_HIP = None


def load_hip():
    """The product library.  Raises if it is not built or cannot be loaded — never falls back to CPU."""
    global _HIP
    if _HIP is None:
        _HIP = OrbxLib(HIP_LIB_PATH)
    return _HIP
