"""Run by tests/test_abi.py in a process of its own: every entry point of include/orbx.h is called with null pointers and zero sizes -
(1) everything null, (2) a live handle that has not extracted anything yet + nulls, (3) a live handle after an extraction + nulls and ones.
A call may succeed (optional outputs) or refuse; it may not dereference.  The name of the call in flight is printed first, so a crash names it."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, sys.argv[2])
from orb_slam3_detailed_comments_amd import _lib, synth                  # noqa: E402
from orb_slam3_detailed_comments_amd.extractor import ORBextractor       # noqa: E402

lib = _lib.OrbxLib(sys.argv[1])
NOT_AN_EXTRACTOR_FIRST = ("orbv_", "orbx_comm_")                          # their first pointer is a vocabulary / a communicator
SKIP_WITH_HANDLE = {"orbx_create", "orbx_destroy", "orbm_points_destroy", "orbm_keyframe_destroy", "orbx_device_count", "orbx_last_error",
                    "orbx_stage_name", "orbx_debug_live_resources"}


def zeros(at, handle, fill):
    args, first = [], True
    for t in at:
        if t in (C.c_float, C.c_double): args.append(0.0)
        elif t in (C.c_int, C.c_size_t, C.c_longlong, C.c_uint, C.c_uint64): args.append(fill)
        elif first and handle is not None and t is C.c_void_p: args.append(handle); first = False
        else: args.append(None)
    return args


calls = 0
for mode in ("null", "fresh", "extracted"):
    ex = None
    if mode != "null":
        ex = ORBextractor(500, 1.2, 8, 20, 7, lib=lib)
        if mode == "extracted":
            ex.extract_batch(np.stack([synth.corner_field(376, 240, seed=s, nrect=750) for s in range(2)]))
    for name in _lib.SYMBOLS:
        f = getattr(lib.L, name)
        if f.argtypes is None:
            continue
        if ex is not None and (name in SKIP_WITH_HANDLE or (name.startswith(NOT_AN_EXTRACTOR_FIRST) and name not in ("orbv_create", "orbv_load_text"))):
            continue
        print(mode, name, flush=True)
        f(*zeros(f.argtypes, ex._h if ex is not None else None, 1 if mode == "extracted" else 0))
        calls += 1
    if ex is not None:
        ex.close()
print("DONE", calls)
