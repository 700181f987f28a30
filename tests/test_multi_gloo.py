"""World-size-2 test of the multi-GPU path on CPU (gloo): streams are sharded rank = s % world with no data-path collective,
each rank's outputs equal the single-process outputs for the same stream, and the optional descriptor all-gather delivers
every rank's fixed-shape block to every rank.  (The kernels run on the CPU SIMT emulator here; on the GPUs the same code
runs with backend nccl = RCCL.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from orb_slam3_detailed_comments_amd import synth, _lib, multi
    from orb_slam3_detailed_comments_amd.extractor import ORBextractor
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _lib.OrbxLib(os.path.join(ROOT, "tests", "emu", "liborbx_emu.so"))
    ex = ORBextractor(300, 1.2, 8, 20, 7, lib=lib)
    frames = [synth.corner_field(376, 240, seed=50 + s, nrect=800) for s in range(5)]     # 5 streams over 2 ranks
    mine = multi.process_streams(ex, frames, rank, world)
    cap = ex.max_keypoints()
    # pad to the same number of streams per rank so that the collective has a fixed shape
    per_rank = (len(frames) + world - 1) // world
    results = [mine[s] for s in sorted(mine)] + [(0, np.zeros(0), np.zeros((0, 32), np.uint8))] * (per_rank - len(mine))
    desc, cnt = multi.pack_descriptors(results, cap)
    dall, call = multi.all_gather_descriptors(desc, cnt)
    # the same exchange straight from the extractor's device-resident blocks (no host packing): every rank extracts per_rank frames so that
    # the blocks have one shape, then one collective per array
    own = [frames[s] for s in sorted(mine)]
    ex.extract_batch(np.stack(own + [own[0]] * (per_rank - len(own))))
    dev_all, dev_cnt, works = multi.all_gather_extracted(ex)
    [w.wait() for w in works]
    dev_all = dev_all.cpu().numpy(); dev_cnt = dev_cnt.cpu().numpy()
    for r in range(world):
        for j, s in enumerate(multi.shard_streams(len(frames), r, world)):
            assert dev_cnt[r, j] == call[r, j] and dev_all[r, j].tobytes() == dall[r, j].tobytes(), "device-resident gather differs from the packed one"
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, sorted(mine), {s: (mine[s][0], mine[s][1].tobytes(), mine[s][2].tobytes()) for s in mine}, dall.tobytes(), call.tolist()))


def test_two_ranks_gloo(emu_lib):
    import torch.multiprocessing as mp
    import oracle_lib as ol
    from orb_slam3_detailed_comments_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [q.get(timeout=300) for _ in procs]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    got.sort()
    assert got[0][1] == [0, 2, 4] and got[1][1] == [1, 3]          # rank = stream % world
    single = {}
    for s in range(5):
        mono, k, d = ol.OracleExtractor(300).extract(synth.corner_field(376, 240, seed=50 + s, nrect=800))
        single[s] = (mono, k.tobytes(), d.tobytes())
    for rank, streams, res, dall, call in got:
        for s in streams:
            assert res[s] == single[s], "stream %d on rank %d differs from the single-process result" % (s, rank)
    assert got[0][3] == got[1][3] and got[0][4] == got[1][4]       # every rank holds the same gathered buffer
    call = np.array(got[0][4])
    assert call.shape == (2, 3) and call[1, 2] == 0                 # rank 1 owns only 2 streams: its third slot is padding
    dall = np.frombuffer(got[0][3], np.uint8).reshape(2, 3, -1, 32)
    for r, streams in ((0, [0, 2, 4]), (1, [1, 3])):
        for j, s in enumerate(streams):
            n = call[r, j]
            assert dall[r, j, :n].tobytes() == single[s][2] and not dall[r, j, n:].any()
