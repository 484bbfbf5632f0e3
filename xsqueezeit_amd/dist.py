"""Multi-GPU sharding of the block path.

Every 8192-line block is independent (fresh PBWT prefix array per block), so contiguous block
ranges go to ranks with no data-path collective during encode or decode.  The one exchange step
is the gather of the compressed block streams to the writer rank: sizes first (tiny all-gather),
then point-to-point sends of exactly each rank's bytes (RCCL grouped send/recv over xGMI).  The cheaper
alternative, exchanging only the sizes and letting every rank pwrite its own byte range, is
exchange_region_offsets + write_own_range.
The functions taking `dist` work on any torch.distributed backend (gloo in the CPU tests).  On the GPUs the
product path is the C ABI (`xsi_hip_shard_blocks`, `xsi_hip_comm_*`, `xsi_hip_gather_block_streams`: RCCL called
from libxsi_hip.so, which is what a C++ host binds); `RcclGather` below drives it from Python and only borrows
torch.distributed to hand the ncclUniqueId to the other ranks.
"""
import ctypes

import numpy as np


def shard_blocks(n_blocks, world_size, rank):
    """Contiguous block range of `rank`: block b goes to rank floor(b*world/n_blocks), so the
    gathered streams concatenate in file order."""
    lo = (rank * n_blocks + world_size - 1) // world_size
    hi = ((rank + 1) * n_blocks + world_size - 1) // world_size
    return lo, hi


def route_queries(bm_positions, n_blocks, world_size):
    """Decode-side sharding (BASELINE configs[4]): the rank that serves each BM position = the owner
    of its block (position >> 15) under shard_blocks.  Returns an int array of ranks; no exchange is involved: every
    rank opens the file (or its block range) read-only and answers the queries routed to it."""
    blocks = np.asarray(bm_positions, dtype=np.int64) >> 15
    # shard_blocks: rank r owns [ceil(r B / G), ceil((r + 1) B / G)), i.e. block b -> rank floor(b G / B)
    return (blocks * world_size // n_blocks).astype(np.int64)


class RcclGather:
    """The writer-rank gather through libxsi_hip.so's own RCCL communicator (include/xsi_hip.h, "multi-GPU").

    ctx: binding.Context of this rank; tdist: an initialised torch.distributed (any backend), used once, to
    broadcast the ncclUniqueId rank 0 makes.  gather() starts the exchange (own stream, behind the encode) and returns
    (region_all, offsets_all, bytes_per_rank, blocks_per_rank) on `dst` - views of buffers this object owns and
    reuses, valid after wait() - and (None, None, bytes_per_rank, blocks_per_rank) elsewhere."""

    def __init__(self, ctx, tdist, device):
        import torch
        from . import binding
        self._b = binding
        self._L = binding.lib()
        self.world = tdist.get_world_size()
        self.rank = tdist.get_rank()
        self.device = device
        idbuf = (ctypes.c_uint8 * 128)()
        if self.rank == 0:
            binding.check(self._L.xsi_hip_comm_unique_id(idbuf))
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=device if tdist.get_backend() == "nccl" else "cpu")
        tdist.broadcast(t, 0)
        idbuf = (ctypes.c_uint8 * 128)(*t.cpu().tolist())
        self.handle = ctypes.c_void_p()
        binding.check(self._L.xsi_hip_comm_create(ctypes.byref(self.handle), ctx.handle, self.world, self.rank, idbuf))
        self._region = self._offs = None

    def gather(self, backing, nbytes, block_offsets_rel, region_capacity, blocks_capacity, dst=0):
        import torch
        if self.rank == dst:
            if self._region is None or self._region.numel() < region_capacity:
                self._region = torch.empty(max(int(region_capacity), 1), dtype=torch.uint8, device=self.device)
            if self._offs is None or self._offs.numel() < blocks_capacity:
                self._offs = torch.empty(max(int(blocks_capacity), 1), dtype=torch.int64, device=self.device)
        offs = block_offsets_rel.to(torch.int64).contiguous()
        per_b = (ctypes.c_uint64 * self.world)()
        per_n = (ctypes.c_uint64 * self.world)()
        on_dst = self.rank == dst
        self._b.check(self._L.xsi_hip_gather_block_streams(
            self.handle, backing.data_ptr(), int(nbytes), offs.data_ptr() if offs.numel() else None, offs.numel(), dst,
            self._region.data_ptr() if on_dst else None, self._region.numel() if on_dst else 0,
            self._offs.data_ptr() if on_dst else None, self._offs.numel() if on_dst else 0, per_b, per_n))
        self._keep = (backing, offs)  # the sends read them until the stream has run
        per_b, per_n = list(per_b), list(per_n)
        if not on_dst:
            return None, None, per_b, per_n
        return self._region[:sum(per_b)], self._offs[:sum(per_n)], per_b, per_n

    def wait(self, host=True):
        """Block until the exchange started last is complete (host=False: make the context's stream wait instead)."""
        self._b.check(self._L.xsi_hip_comm_wait(self.handle, 1 if host else 0))
        self._keep = None

    def close(self):
        if self.handle:
            self._L.xsi_hip_comm_destroy(self.handle)
            self.handle = None


class _GatherHandle:
    """Result of gather_block_streams_async: wait() -> (parts, offsets) on the writer rank, None elsewhere."""

    def __init__(self, works, keep, finish):
        self._works = works
        self._keep = keep      # tensors the collectives still read
        self._finish = finish

    def wait(self):
        for w in self._works:
            w.wait()
        self._keep = None
        return self._finish()


_RECV_CACHE = {}


def gather_block_streams_async(backing, nbytes, block_offsets, dist, device=None, dst=0):
    """Start the gather of this rank's blocks region to rank `dst` and return a handle.

    backing: 1-D uint8 tensor whose first `nbytes` bytes are this rank's blocks region (the encode output
    buffer: it is sent from in place, no staging copy); block_offsets: 1-D int64 tensor, offsets relative
    to the region start.
    handle.wait() returns on `dst` (parts, offsets): per-rank uint8 views in rank order (= file order)
    and per-rank int64 offsets relative to the start of the concatenated region; None elsewhere.
    One size all-gather, then point-to-point sends of exactly the bytes each rank has (no padding to the
    longest region), issued asynchronously so the caller can overlap them with work that only reads
    `backing`.  The receive buffers are reused across calls; the writer rank's own part is a VIEW of its
    `backing` (no copy): consume or copy it before the next encode overwrites that buffer."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    device = device if device is not None else backing.device
    meta = torch.tensor([int(nbytes), block_offsets.numel()], dtype=torch.int64, device=device)
    metas = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, meta)
    sizes = [int(m[0]) for m in metas]
    nblk = [int(m[1]) for m in metas]
    if backing.numel() < sizes[rank]:
        raise ValueError("backing buffer (%d bytes) shorter than its region (%d)" % (backing.numel(), sizes[rank]))
    if sizes[rank] == 0 and backing.numel() == 0:
        backing = torch.zeros(1, dtype=torch.uint8, device=device)  # a rank without blocks still posts a 1-byte send
    send = backing[:max(sizes[rank], 1)]
    offs = block_offsets.to(torch.int64).contiguous() if block_offsets.numel() else torch.zeros(1, dtype=torch.int64, device=device)
    bufs = obufs = None
    ops = []
    if rank == dst:
        key = (str(device), world)
        cached = _RECV_CACHE.get(key)
        if cached is None or any(cached[0][r].numel() < max(sizes[r], 1) or cached[1][r].numel() < max(nblk[r], 1)
                                 for r in range(world)):
            cached = ([torch.empty(max(sizes[r], 1), dtype=torch.uint8, device=device) for r in range(world)],
                      [torch.empty(max(nblk[r], 1), dtype=torch.int64, device=device) for r in range(world)])
            _RECV_CACHE[key] = cached
        bufs = [cached[0][r][:max(sizes[r], 1)] for r in range(world)]
        obufs = [cached[1][r][:max(nblk[r], 1)] for r in range(world)]
        for r in range(world):
            if r == dst:
                continue
            ops.append(dist.P2POp(dist.irecv, bufs[r], r))
            ops.append(dist.P2POp(dist.irecv, obufs[r], r))
    else:
        ops.append(dist.P2POp(dist.isend, send, dst))
        ops.append(dist.P2POp(dist.isend, offs, dst))
    works = dist.batch_isend_irecv(ops) if ops else []

    def finish():
        if rank != dst:
            return None
        parts, out_offs = [], []
        base = 0
        for r in range(world):
            part = send[:sizes[r]] if r == dst else bufs[r][:sizes[r]]
            o = offs[:nblk[r]] if r == dst else obufs[r][:nblk[r]]
            parts.append(part)
            out_offs.append(o + base)
            base += sizes[r]
        return parts, out_offs

    return _GatherHandle(works, (send, offs, bufs, obufs), finish)


def exchange_region_offsets(nbytes, dist, device=None):
    """The cheaper alternative to the gather: exchange only the region SIZES; every rank
    then writes its own byte range of the output file itself.  Returns (my_start, total) in bytes of the
    concatenated blocks region (file offset = 256 + my_start)."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    meta = torch.tensor([int(nbytes)], dtype=torch.int64, device=device)
    metas = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, meta)
    sizes = [int(m[0]) for m in metas]
    return sum(sizes[:rank]), sum(sizes)


def write_own_range(path, region_bytes, my_start):
    """pwrite this rank's blocks region at its place in the shared output file (created by the writer rank
    with the header; ranges of different ranks do not overlap, so no ordering between ranks is needed)."""
    import os
    fd = os.open(path, os.O_WRONLY)
    try:
        view = memoryview(region_bytes)
        done = 0
        while done < len(view):
            done += os.pwrite(fd, view[done:], 256 + my_start + done)
    finally:
        os.close(fd)


def gather_block_streams(region, block_offsets, dist, device=None, dst=0):
    """Blocking form: region is exactly this rank's blocks region.  Returns on rank `dst`
    (region_all uint8 tensor, offsets_all int64 tensor relative to the start of the concatenated
    region); None elsewhere."""
    import torch
    device = device if device is not None else region.device
    backing = region if region.numel() else torch.zeros(1, dtype=torch.uint8, device=device)
    out = gather_block_streams_async(backing, region.numel(), block_offsets, dist, device, dst).wait()
    if out is None:
        return None
    parts, offs = out
    return torch.cat(parts), torch.cat(offs)


def assemble_file(region_bytes, offsets_rel, header_fields_fn, sample_names):
    """Host-side tail of XsiFactoryExt::finalize_file (xsi_factory.hpp:558-605): pad the blocks
    region to 8, append the u64 index (file offsets = 256 + relative offset) and the sample names,
    and prepend the header made by header_fields_fn(indices_offset, samples_offset) -> 256 bytes."""
    body = bytearray(region_bytes)
    while (256 + len(body)) % 8:
        body.append(0)
    indices_offset = 256 + len(body)
    body += (np.asarray(offsets_rel, dtype="<u8") + np.uint64(256)).astype("<u8").tobytes()
    samples_offset = 256 + len(body)
    for s in sample_names:
        body += s.encode() + b"\0"
    return bytes(header_fields_fn(indices_offset, samples_offset)) + bytes(body)
