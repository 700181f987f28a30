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


def extracted_blocks(extractor):
    """A snapshot of the extractor's device-resident results of the last batch as tensors the CALLER owns: (desc [B, cap, 32] u8 with zero
    rows beyond the count, counts [B] i32), on the extractor's own GPU (orbx_device_id - not torch's current device).  The copy is device to
    device on the handle's stream behind the extraction and complete on return (orbx_device_snapshot), so the handle may start its next
    extraction - which clears and rewrites its descriptor buffer - while a collective still reads these tensors.  (The library of the CPU tests
    reports ORBX_DEVICE_HOST: its "device" memory is host memory and the tensors are CPU tensors.)"""
    import ctypes as C
    import torch
    L = extractor._lib
    cap = C.c_int(); B = C.c_int()
    L.check(L.L.orbx_device_outputs(extractor._h, None, None, None, None, C.byref(cap), C.byref(B)))
    dev_id = L.L.orbx_device_id(extractor._h)
    dev = torch.device("cpu") if dev_id < 0 else torch.device("cuda", dev_id)
    d = torch.empty((B.value, cap.value, 32), dtype=torch.uint8, device=dev)
    c = torch.empty((B.value,), dtype=torch.int32, device=dev)
    if dev_id >= 0:
        torch.cuda.current_stream(dev).synchronize()        # the allocator may hand out memory that work queued on torch's stream still uses
    L.check(L.L.orbx_device_snapshot(extractor._h, d.data_ptr(), c.data_ptr()))
    return d, c


def all_gather_extracted(extractor, side_stream=None):
    """All-gather of the descriptor blocks of the last batch from the device: a device-to-device snapshot (extracted_blocks), then one collective
    for the descriptors and one for the counts (RCCL over xGMI with backend "nccl"), no host bounce.  Returns (desc_all [world, B, cap, 32],
    counts_all [world, B]) as tensors on the extractor's device and the work handles; with `side_stream` (a torch.cuda.Stream) the collectives
    are issued there with async_op=True so that the next extraction - which runs on the handle's own HIP streams and rewrites the handle's
    buffers - overlaps them; call .wait() on the handles before reading."""
    import contextlib
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    d, c = extracted_blocks(extractor)
    B = d.shape[0]
    dall = torch.empty((world * B,) + tuple(d.shape[1:]), dtype=d.dtype, device=d.device); call = torch.empty((world * B,), dtype=c.dtype, device=c.device)
    ctx = torch.cuda.stream(side_stream) if (side_stream is not None and d.is_cuda) else contextlib.nullcontext()
    with ctx:
        if side_stream is not None and d.is_cuda:
            d.record_stream(side_stream); c.record_stream(side_stream); dall.record_stream(side_stream); call.record_stream(side_stream)
        w1 = dist.all_gather_into_tensor(dall, d, async_op=True)      # rank r's block lands at rows [r * B, (r + 1) * B)
        w2 = dist.all_gather_into_tensor(call, c, async_op=True)
    return dall.view((world, B) + tuple(d.shape[1:])), call.view(world, B), (w1, w2)


class Communicator:
    """The exchange step behind the C ABI (include/orbx.h: orbx_comm_*, orbx_allgather_descriptors): RCCL over xGMI inside the library, no torch.
    A C++ host calls the same entry points.  `unique_id` = Communicator.unique_id(lib) on rank 0, handed to the other ranks by any means
    (from_torch_distributed below uses the process group that launched the ranks)."""

    def __init__(self, lib, world, rank, unique_id, device_id=0):
        import ctypes as C
        self._lib = lib
        self.world, self.rank = int(world), int(rank)
        idb = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        h = C.c_void_p()
        lib.check(lib.L.orbx_comm_create(C.byref(h), self.world, self.rank, idb, int(device_id)))
        self._c = h
        self.B = self.cap = 0

    @staticmethod
    def unique_id(lib):
        import ctypes as C
        idb = (C.c_uint8 * 128)()
        lib.check(lib.L.orbx_comm_unique_id(idb))
        return bytes(idb)

    @classmethod
    def from_torch_distributed(cls, lib, device_id=0):
        """One communicator over the ranks of the default torch.distributed group: rank 0's unique id travels through broadcast_object_list
        (any backend; gloo in the CPU tests), the collectives themselves run inside the library."""
        import torch.distributed as dist
        box = [cls.unique_id(lib) if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(lib, dist.get_world_size(), dist.get_rank(), box[0], device_id)

    def all_gather(self, extractor):
        """Enqueue the all-gather of the descriptor blocks + counts of `extractor`'s last batch (asynchronous: the extractor may start its next batch)."""
        import ctypes as C
        B = C.c_int(); cap = C.c_int(); d = C.c_void_p(); n = C.c_void_p()
        self._lib.check(self._lib.L.orbx_allgather_descriptors(extractor._h, self._c, C.byref(d), C.byref(n), C.byref(B), C.byref(cap)))
        self.B, self.cap, self.desc_ptr, self.n_ptr = B.value, cap.value, d, n

    def wait(self):
        self._lib.check(self._lib.L.orbx_comm_wait(self._c))

    def fetch(self):
        """(desc_all [world, B, cap, 32] u8, n_all [world, B] i32) on the host; waits for the exchange."""
        desc = np.zeros((self.world, self.B, self.cap, 32), np.uint8); n = np.zeros((self.world, self.B), np.int32)
        self._lib.check(self._lib.L.orbx_comm_fetch(self._c, desc.ctypes.data, n.ctypes.data))
        return desc, n

    def close(self):
        if self._c:
            self._lib.L.orbx_comm_destroy(self._c); self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def process_streams(extractor, frames_of_stream, rank, world, lap=(0, 0)):
    """Extracts one frame batch: frames_of_stream[s] is the current image of stream s (all the same size); this rank
    processes its own streams in one batched launch.  Returns {stream: (mono, kps, desc)}."""
    mine = shard_streams(len(frames_of_stream), rank, world)
    if not mine:
        return {}
    res = extractor.extract_batch(np.stack([frames_of_stream[s] for s in mine]), lap)
    return dict(zip(mine, res))
