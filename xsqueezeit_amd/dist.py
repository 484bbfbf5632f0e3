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


def gather_block_streams(region, block_offsets, dist, device=None, dst=0):
    """region: 1-D uint8 tensor holding this rank's blocks region (blocks padded to 4 bytes);
    block_offsets: 1-D int64 tensor of this rank's block offsets relative to ITS region start.
    Returns on rank `dst` (region_all uint8 tensor, offsets_all int64 tensor relative to the start
    of the concatenated region); None elsewhere.  One size all-gather + one payload gather."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    device = device if device is not None else region.device
    meta = torch.tensor([region.numel(), block_offsets.numel()], dtype=torch.int64, device=device)
    metas = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, meta)
    sizes = [int(m[0]) for m in metas]
    nblk = [int(m[1]) for m in metas]
    max_sz = max(max(sizes), 1)
    max_nb = max(max(nblk), 1)
    pad = torch.zeros(max_sz, dtype=torch.uint8, device=device)
    pad[:region.numel()] = region
    offp = torch.zeros(max_nb, dtype=torch.int64, device=device)
    offp[:block_offsets.numel()] = block_offsets
    if rank == dst:
        bufs = [torch.zeros(max_sz, dtype=torch.uint8, device=device) for _ in range(world)]
        obufs = [torch.zeros(max_nb, dtype=torch.int64, device=device) for _ in range(world)]
    else:
        bufs = obufs = None
    dist.gather(pad, bufs, dst=dst)
    dist.gather(offp, obufs, dst=dst)
    if rank != dst:
        return None
    parts, offs = [], []
    base = 0
    for r in range(world):
        parts.append(bufs[r][:sizes[r]])
        offs.append(obufs[r][:nblk[r]] + base)
        base += sizes[r]
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
