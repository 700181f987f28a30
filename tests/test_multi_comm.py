"""The exchange step of the multi-GPU mode through the C ABI (orbx_comm_*, orbx_allgather_descriptors; BASELINE.json configs[4]).
CPU: two "ranks" = two threads of this process, each with an extractor on its own "GPU" of the emulator build (ORBX_EMU_DEVICES=2), meeting in
the emulator's in-process rendezvous - entry points, layouts and the overlap contract (the handle extracts its next batch while the gathered
blocks stay intact) are the product's; RCCL itself is exercised on the GPU box with a one-rank communicator (ncclCommInitRank, ncclAllGather)."""
import os
import threading

import numpy as np
import pytest

from orb_slam3_detailed_comments_amd import ORBextractor, synth, multi, OrbxError


def _images(rank, B, w, h):
    return np.stack([synth.corner_field(w, h, seed=900 + 17 * rank + b, nrect=max(200, w * h // 200)) for b in range(B)])


def test_allgather_two_ranks_emulated(emu_lib, monkeypatch):
    monkeypatch.setenv("ORBX_EMU_DEVICES", "2")
    assert emu_lib.L.orbx_device_count() == 2
    uid = multi.Communicator.unique_id(emu_lib)
    world, B, w, h = 2, 3, 320, 240
    out, err = [None] * world, []

    def rank_main(rank):
        try:
            ex = ORBextractor(300, 1.2, 8, 20, 7, device_id=rank, lib=emu_lib)        # every rank on its own device
            comm = multi.Communicator(emu_lib, world, rank, uid, device_id=rank)
            res = ex.extract_batch(_images(rank, B, w, h))
            comm.all_gather(ex)
            res2 = ex.extract_batch(_images(rank + 5, B, w, h))                        # the next batch overwrites the handle's buffers meanwhile
            comm.wait()
            d, n = comm.fetch()
            comm.all_gather(ex)                                                         # a second round on the same communicator
            d2, n2 = comm.fetch()
            out[rank] = (res, d, n, res2, d2, n2)
            comm.close(); ex.close()
        except Exception as e:                                                          # pragma: no cover
            err.append(e)
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(120) for t in th]
    assert not err, err
    for rank in range(world):
        for first, (ri, di, ni) in enumerate(((0, 1, 2), (3, 4, 5))):
            d, n = out[rank][di], out[rank][ni]
            assert d.shape[:2] == (world, B)
            for src in range(world):
                res = out[src][ri]
                for b in range(B):
                    k = len(res[b][2])
                    assert n[src, b] == k and np.array_equal(d[src, b, :k], res[b][2]) and not d[src, b, k:].any(), (rank, src, b)
    # ranks on different devices cannot share a communicator slot: the extractor's device must be the communicator's
    ex0 = ORBextractor(300, 1.2, 8, 20, 7, device_id=0, lib=emu_lib)
    ex0.extract_batch(_images(0, 1, w, h))
    c1 = multi.Communicator(emu_lib, 1, 0, multi.Communicator.unique_id(emu_lib), device_id=1)
    with pytest.raises(OrbxError):
        c1.all_gather(ex0)
    c1.close(); ex0.close()


@pytest.mark.gpu
def test_allgather_one_rank_rccl_gpu(hip_lib):
    """RCCL through the library (dlopen'ed librccl: ncclGetUniqueId, ncclCommInitRank, grouped ncclAllGather x2 on the communicator's stream)."""
    B, w, h = 4, 752, 480
    ex = ORBextractor(1200, 1.2, 8, 20, 7, device_id=0, lib=hip_lib)
    comm = multi.Communicator(hip_lib, 1, 0, multi.Communicator.unique_id(hip_lib), device_id=0)
    res = ex.extract_batch(_images(0, B, w, h))
    comm.all_gather(ex)
    ex.extract_batch(_images(3, B, w, h))
    d, n = comm.fetch()
    for b in range(B):
        k = len(res[b][2])
        assert n[0, b] == k and np.array_equal(d[0, b, :k], res[b][2]) and not d[0, b, k:].any()
    comm.close(); ex.close()


_ORDER_PROBE = r'''
import ctypes as C, os, sys
order, root, want_exchange = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
sys.path.insert(0, root)
if order == "torch_first":
    import torch
from orb_slam3_detailed_comments_amd import _lib, multi
lib = _lib.load_hip()
def mapped(key):
    return sorted(set(l.split()[-1] for l in open("/proc/self/maps") if key in l))
hip_of_library = [p for p in mapped("libamdhip64")]
if order == "lib_first":
    import torch
idb = (C.c_uint8 * 128)()
rc = lib.L.orbx_comm_unique_id(idb)                 # loads RCCL
print("ORBX_HIP=%r" % (hip_of_library,)); print("ORBX_RCCL=%r" % (mapped("librccl"),)); print("ORBX_RC=%d" % rc)
if want_exchange:
    import numpy as np
    from orb_slam3_detailed_comments_amd import synth
    from orb_slam3_detailed_comments_amd.extractor import ORBextractor
    ex = ORBextractor(1200, 1.2, 8, 20, 7, device_id=0, lib=lib)
    res = ex.extract_batch(np.stack([synth.corner_field(752, 480, seed=s) for s in range(2)]))
    comm = multi.Communicator(lib, 1, 0, multi.Communicator.unique_id(lib), device_id=0)
    comm.all_gather(ex); d, n = comm.fetch()
    assert all(n[0, b] == len(res[b][2]) and np.array_equal(d[0, b, :n[0, b]], res[b][2]) for b in range(2))
    comm.close(); ex.close()
    print("EXCHANGE OK")
'''


def _order_probe(order, exchange):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _ORDER_PROBE, order, root, "1" if exchange else "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = dict(l.split("=", 1) for l in r.stdout.splitlines() if l.startswith("ORBX_"))       # (RCCL prints a banner of its own on stdout)
    return eval(out["ORBX_HIP"]), eval(out["ORBX_RCCL"]), r.stdout


def _check_same_directory(hip, rccl):
    # the library resolved exactly one HIP runtime, and an RCCL from that runtime's directory is mapped (the one it talks to)
    assert len(hip) == 1, hip
    d = os.path.dirname(os.path.realpath(hip[0]))
    assert any(os.path.dirname(os.path.realpath(p)) == d for p in rccl), (hip, rccl)


def test_rccl_is_taken_from_the_hip_runtime_the_library_is_bound_to():
    """A process can hold two HIP runtimes: this library on /opt/rocm's, PyTorch (imported later) on the copy bundled in torch/lib, which brings its
    own librccl.so.1.  A dlopen by soname would hand the library torch's RCCL - on the other runtime (found on the GPU by tests/test_lifetime.py:
    ncclCommInitRank "unhandled cuda error").  The loader therefore looks next to the libamdhip64 it resolved; both load orders are checked."""
    lib_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orb_slam3_detailed_comments_amd", "liborbx_hip.so")
    if not os.path.exists(lib_path):
        pytest.skip("liborbx_hip.so not built")
    for order in ("lib_first", "torch_first"):
        hip, rccl, _ = _order_probe(order, False)
        _check_same_directory(hip, rccl)


@pytest.mark.gpu
def test_allgather_one_rank_rccl_in_both_load_orders_gpu(hip_lib):
    for order in ("lib_first", "torch_first"):
        hip, rccl, out = _order_probe(order, True)
        _check_same_directory(hip, rccl)
        assert "EXCHANGE OK" in out
