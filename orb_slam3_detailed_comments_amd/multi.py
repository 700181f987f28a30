"""Multi-GPU layer: the path shards over independent image streams — one process per GPU, stream s on rank
s % world, no collective on the data path.  The only exchange is the OPTIONAL all-gather of fixed-shape descriptor blocks
([B, cap, 32] u8 + [B] counts) for a global matcher; it goes through torch.distributed (backend "nccl" = RCCL over xGMI on the
GPUs, "gloo" in the CPU tests)."""
import numpy as np


def shard_streams(n_streams, rank, world):
    """Streams owned by `rank`: s with s % world == rank (config 5 of BASELINE.json: MH01-05, V1_01-03 over 8 GPUs)."""
    return [s for s in range(n_streams) if s % world == rank]


def pack_descriptors(results, cap):
    """results: list of (mono, kps, desc) per image -> (desc [B,cap,32] u8 zero-padded, counts [B] i32)."""
    B = len(results)
    out = np.zeros((B, cap, 32), np.uint8); cnt = np.zeros(B, np.int32)
    for b, (_, _, d) in enumerate(results):
        out[b, :len(d)] = d; cnt[b] = len(d)
    return out, cnt


def all_gather_descriptors(desc, counts, device=None):
    """All-gather one fixed-shape block per rank.  Returns (desc_all [world,B,cap,32], counts_all [world,B]).
    One collective per frame batch (latency-bound at these sizes: ~38 KB per frame)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    d = torch.from_numpy(np.ascontiguousarray(desc)); c = torch.from_numpy(np.ascontiguousarray(counts))
    if device is not None:
        d = d.to(device, non_blocking=True); c = c.to(device, non_blocking=True)
    dl = [torch.empty_like(d) for _ in range(world)]; cl = [torch.empty_like(c) for _ in range(world)]
    dist.all_gather(dl, d); dist.all_gather(cl, c)
    return torch.stack(dl).cpu().numpy(), torch.stack(cl).cpu().numpy()


def process_streams(extractor, frames_of_stream, rank, world, lap=(0, 0)):
    """Extracts one frame batch: frames_of_stream[s] is the current image of stream s (all the same size); this rank
    processes its own streams in one batched launch.  Returns {stream: (mono, kps, desc)}."""
    mine = shard_streams(len(frames_of_stream), rank, world)
    if not mine:
        return {}
    res = extractor.extract_batch(np.stack([frames_of_stream[s] for s in mine]), lap)
    return dict(zip(mine, res))
