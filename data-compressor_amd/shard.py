"""Multi-PROCESS sharding helpers (one process per GPU, the layout bench.py is launched in): channels are independent
(every stream starts from last_value = 0 and a fresh model: DCLib/src/diff.c:11, bac.c:150), so N GPUs = N disjoint channel
ranges and NO data-path collective.  Inside ONE process the product does all of this itself (dega_hip_group_* of
include/dega_hip.h: one context and one host thread per device, host-side concatenate); when every GPU has a process of its
own, what remains is the channel partition and bringing the ranks' packed streams together on one HOST: gather_streams
moves the streams to host memory first and gathers them there (gloo group) -- never a device-side collective."""


def channel_range(rank, world, total_channels):
    """Contiguous range [c0, c1) of rank `rank`; the first (total % world) ranks get one channel more."""
    base, extra = divmod(total_channels, world)
    c0 = rank * base + min(rank, extra)
    return c0, c0 + base + (1 if rank < extra else 0)


def pack_streams(streams, bits):
    """[C, cap] slabs + bit lengths (numpy) -> (packed bytes, byte sizes): each stream padded to a whole byte."""
    import numpy as np
    sizes = ((np.asarray(bits, dtype=np.uint64) + np.uint64(7)) // np.uint64(8)).astype(np.int64)
    packed = np.concatenate([streams[c, : sizes[c]] for c in range(streams.shape[0])]) if streams.shape[0] else np.zeros(0, dtype=np.uint8)
    return packed.astype(np.uint8, copy=False), sizes


def gather_streams(packed, bits, group=None, dst=0):
    """Collect every rank's (packed bytes, bit lengths) on `dst`'s HOST in rank (= channel) order.
    packed: 1-D uint8 tensor, bits: 1-D int64 tensor; device tensors are copied to host memory first, and `group` must be
    a process group that carries host tensors (gloo): the streams travel host to host, no RCCL / xGMI traffic.
    Returns (packed_all, bits_all, offsets) as host tensors on dst -- offsets[c] = first byte of channel c -- and
    (None, None, None) elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if "gloo" not in str(dist.get_backend(group)).lower():
        raise RuntimeError("gather_streams concatenates on the host: pass a gloo process group (dist.new_group(backend='gloo'))")
    packed, bits = packed.cpu(), bits.cpu()
    dev = packed.device
    meta = torch.tensor([packed.numel(), bits.numel()], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    max_bytes = max(int(m[0]) for m in metas)
    max_ch = max(int(m[1]) for m in metas)
    pad_p = torch.zeros(max(max_bytes, 1), dtype=torch.uint8, device=dev)
    pad_p[: packed.numel()] = packed
    pad_b = torch.zeros(max(max_ch, 1), dtype=torch.int64, device=dev)
    pad_b[: bits.numel()] = bits
    got_p = [torch.zeros_like(pad_p) for _ in range(world)] if rank == dst else None
    got_b = [torch.zeros_like(pad_b) for _ in range(world)] if rank == dst else None
    dist.gather(pad_p, got_p, dst=dst, group=group)
    dist.gather(pad_b, got_b, dst=dst, group=group)
    if rank != dst:
        return None, None, None
    packed_all = torch.cat([got_p[r][: int(metas[r][0])] for r in range(world)])
    bits_all = torch.cat([got_b[r][: int(metas[r][1])] for r in range(world)])
    sizes = (bits_all + 7) // 8
    offsets = torch.zeros(bits_all.numel() + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(sizes, 0)
    return packed_all, bits_all, offsets
