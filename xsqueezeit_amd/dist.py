"""Multi-GPU sharding of the block path.

Every 8192-line block is independent (fresh PBWT prefix array per block), so contiguous block
ranges go to ranks with no data-path collective during encode or decode.  The one exchange step
is the gather of the compressed block streams to the writer rank: sizes first (tiny all-gather),
then the variable-length streams, padded to the longest, in one RCCL all-gather-style gather.
Works on any torch.distributed backend (RCCL on the GPUs, gloo in the CPU tests).
"""
import numpy as np


def shard_blocks(n_blocks, world_size, rank):
    """Contiguous block range of `rank`: block b goes to rank floor(b*world/n_blocks), so the
    gathered streams concatenate in file order."""
    lo = (rank * n_blocks + world_size - 1) // world_size
    hi = ((rank + 1) * n_blocks + world_size - 1) // world_size
    return lo, hi


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

    backing: 1-D uint8 tensor whose first `nbytes` bytes are this rank's blocks region and whose
    capacity is at least the longest region of any rank (the encode output buffer: the send is padded
    to the longest region by sending that many bytes of it, so no staging copy is made);
    block_offsets: 1-D int64 tensor, offsets relative to the region start.
    handle.wait() returns on `dst` (parts, offsets): per-rank uint8 views in rank order (= file order)
    and per-rank int64 offsets relative to the start of the concatenated region; None elsewhere.
    The receive buffers are reused across calls.  One size all-gather + two gathers, issued
    asynchronously so the caller can overlap them with work that only reads `backing`."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    device = device if device is not None else backing.device
    meta = torch.tensor([int(nbytes), block_offsets.numel()], dtype=torch.int64, device=device)
    metas = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, meta)
    sizes = [int(m[0]) for m in metas]
    nblk = [int(m[1]) for m in metas]
    max_sz = max(max(sizes), 1)
    max_nb = max(max(nblk), 1)
    if backing.numel() < max_sz:
        raise ValueError("backing buffer (%d bytes) shorter than the longest region (%d)" % (backing.numel(), max_sz))
    send = backing[:max_sz]
    offp = torch.zeros(max_nb, dtype=torch.int64, device=device)
    offp[:block_offsets.numel()] = block_offsets
    bufs = obufs = None
    if rank == dst:
        key = (str(device), world)
        cached = _RECV_CACHE.get(key)
        if cached is None or cached[0][0].numel() < max_sz or cached[1][0].numel() < max_nb:
            cached = ([torch.empty(max_sz, dtype=torch.uint8, device=device) for _ in range(world)],
                      [torch.empty(max_nb, dtype=torch.int64, device=device) for _ in range(world)])
            _RECV_CACHE[key] = cached
        bufs = [b[:max_sz] for b in cached[0]]
        obufs = [b[:max_nb] for b in cached[1]]
    works = [dist.gather(send, bufs, dst=dst, async_op=True), dist.gather(offp, obufs, dst=dst, async_op=True)]

    def finish():
        if rank != dst:
            return None
        parts, offs = [], []
        base = 0
        for r in range(world):
            parts.append(bufs[r][:sizes[r]])
            offs.append(obufs[r][:nblk[r]] + base)
            base += sizes[r]
        return parts, offs

    return _GatherHandle(works, (send, offp, bufs, obufs), finish)


def gather_block_streams(region, block_offsets, dist, device=None, dst=0):
    """Blocking form: region is exactly this rank's blocks region.  Returns on rank `dst`
    (region_all uint8 tensor, offsets_all int64 tensor relative to the start of the concatenated
    region); None elsewhere."""
    import torch
    world = dist.get_world_size()
    device = device if device is not None else region.device
    # every rank needs a send buffer as long as the longest region: learn it, then pad
    meta = torch.tensor([region.numel()], dtype=torch.int64, device=device)
    metas = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_sz = max(max(int(m[0]) for m in metas), 1)
    backing = torch.zeros(max_sz, dtype=torch.uint8, device=device)
    backing[:region.numel()] = region
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
