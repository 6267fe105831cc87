"""Multi-GPU helpers: one process per GPU, rays shard across ranks, ONE collective per step.

The reference is data-parallel only (DDP gradient all-reduce, code/training/monosdf_train.py:228-229;
no DistributedSampler, so every rank draws its own ray batch).  Here the same semantics are spelled
out so the benchmark and custom loops do not need DDP: average the flat gradient with a single
all-reduce (2.7 MB for the 8x256 network -- one message instead of DDP's buckets; on xGMI a ring
all-reduce of that size is latency-bound, so fewer, larger messages are the right shape)."""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_slice(n_items, rank_=None, world_=None):
    """Contiguous, balanced [start, stop) of this rank's share of n_items rays / chunks."""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    base, rem = divmod(n_items, w)
    start = r * base + min(r, rem)
    return start, start + base + (1 if r < rem else 0)


def average_gradients(params):
    """In-place mean of .grad over all ranks with one flat all-reduce."""
    w = world()
    if w == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat /= w
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def all_gather_rows(t):
    """Concatenate per-rank row blocks (rendered pixels of an image sharded by rays), ragged allowed."""
    w = world()
    if w == 1:
        return t
    n = torch.tensor([t.shape[0]], device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    pad[:t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], 0)
