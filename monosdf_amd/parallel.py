"""Multi-GPU helpers: one process per GPU, rays shard across ranks.

The reference is data-parallel only (DDP gradient all-reduce, code/training/monosdf_train.py:228-229;
no DistributedSampler, so every rank draws its own ray batch).  Here the same semantics are spelled
out so the benchmark and custom loops do not need DDP: the MLP gradients travel as ONE flat all-reduce
(2.7 MB for the 8x256 network -- one message instead of DDP's buckets; on xGMI a ring all-reduce of that
size is latency-bound, so fewer, larger messages are the right shape), the hash-grid table's 48.8 MB
gradient as a second one that starts behind the scatter kernel and runs under the weight-gradient kernels."""
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


# ---------------------------------------------------------------------------
# "this gradient is complete" events: the kernel that writes a large gradient (the hash-grid scatter) marks the
# tensor right behind its launch, so that the exchange of that tensor can start while the rest of the backward
# pass (the weight-gradient kernels) is still running -- the comm stream waits for the mark, not for the stream.
# ---------------------------------------------------------------------------
_LISTENERS = 0          # GradientAverager objects with an overlapped block (nobody listening: no event is recorded)
_GRAD_READY = {}        # data_ptr of the gradient tensor -> (tensor, its version counter, event recorded behind the
                        # kernel that completed it)


def mark_grad_ready(t):
    """Called by ops right after the launch that completes gradient tensor ``t`` (on the current stream)."""
    if _LISTENERS > 0 and t.is_cuda:
        if len(_GRAD_READY) > 16:
            _GRAD_READY.clear()
        ev = torch.cuda.Event()
        ev.record()
        _GRAD_READY[t.data_ptr()] = (t, t._version, ev)


def grad_ready_event(g):
    """The event behind which gradient tensor ``g`` is complete, or None when that is not known.  Only for the very
    tensor that was marked and only while nothing has written to it since: when autograd ADDS a second contribution
    into the marked tensor in place (two nodes feeding one table in the same backward pass) the address is the same
    but the version counter has moved, and when it clones or accumulates into another tensor an old entry may sit at
    a recycled address -- in both cases the caller has to wait for the stream instead."""
    if not g.is_cuda:
        return None
    entry = _GRAD_READY.pop(g.data_ptr(), None)
    if entry is None:
        return None
    t, version, ev = entry
    if t is g and g._version == version:
        return ev
    return None


class GradientAverager:
    """Mean of .grad over all ranks, as TWO blocks (the reference: DDP's bucketed all-reduce overlapped with the
    backward pass, training/monosdf_train.py:228-229):

    * the large parameters (``numel >= overlap_min_numel``: the hash-grid embedding table, 48.8 MB) are reduced
      one by one, each as soon as autograd has accumulated its gradient (a post-accumulate hook), on a comm
      stream that waits only for the kernel that completed that gradient (``mark_grad_ready``): the scatter is
      ordered before the SDF weight-gradient launch in the node's backward, so the exchange runs under the
      weight-gradient kernels.  On xGMI 48.8 MB is 0.08 ms as a direct reduce-scatter + all-gather over the 7
      links and up to 0.57 ms as a ring bound by one link: 2-14 % of the 4.2 ms grid step
      if left exposed;
    * everything else (1.2 MB with the hash grid, 2.7 MB for the 8x256 network) is ONE all-reduce of a
      persistent flat buffer in ``average()``; afterwards each ``p.grad`` IS a view of the buffer (no copy back,
      no per-step allocation).  Its layout is fixed at construction from the parameters that require gradients,
      so every rank reduces the same message even when a parameter has no gradient on some rank.

    Every rank must call ``average()`` once per backward pass, and must produce gradients for the same large
    parameters (a large parameter without a gradient is reduced as zeros inside ``average()``, before the flat
    block, which keeps the order of collectives equal as long as there is one such parameter or all ranks miss
    the same ones)."""

    def __init__(self, params, group=None, overlap=True, overlap_min_numel=1 << 20, timing=False):
        global _LISTENERS
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no parameter requires a gradient')
        self.group = group
        self.big = [p for p in self.params if overlap and p.numel() >= overlap_min_numel]
        big_ids = {id(p) for p in self.big}
        self.small = [p for p in self.params if id(p) not in big_ids]
        self.flat, self.views = None, []
        if self.small:
            dev, dt = self.small[0].device, self.small[0].dtype
            self.flat = torch.zeros(sum(p.numel() for p in self.small), device=dev, dtype=dt)
            off = 0
            for p in self.small:
                self.views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        self.timing = timing
        self.timings = {'flat': [], 'overlapped': []}      # (start, end) event pairs when timing is on
        self._inflight = {}                                # id(p) -> (work, done event or None)
        self._comm = None
        self._handles = [p.register_post_accumulate_grad_hook(self._grad_arrived) for p in self.big]
        if self.big:
            _LISTENERS += 1
        self._listening = bool(self.big)

    def close(self):
        global _LISTENERS
        for h in self._handles:
            h.remove()
        self._handles = []
        if self._listening:
            _LISTENERS -= 1
            self._listening = False

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _avg_op(self):
        # RCCL averages in the collective; gloo only sums (the division follows in average())
        return dist.ReduceOp.AVG if dist.get_backend(self.group) == 'nccl' else dist.ReduceOp.SUM

    def _reduce_big(self, p, ready=None):
        g = p.grad
        op = self._avg_op()
        if not g.is_cuda:
            self._inflight[id(p)] = (dist.all_reduce(g, op=op, group=self.group, async_op=True), None, op)
            return
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=g.device)
        comm, main = self._comm, torch.cuda.current_stream(g.device)
        if ready is not None:
            comm.wait_event(ready)         # the kernel that completed this gradient, not what was enqueued since
        else:
            comm.wait_stream(main)
        with torch.cuda.stream(comm):
            g.record_stream(comm)
            if self.timing:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(comm)
            work = dist.all_reduce(g, op=op, group=self.group, async_op=True)
            work.wait()                    # orders the comm stream behind the collective (no host wait on RCCL)
            done = torch.cuda.Event(enable_timing=self.timing)
            done.record(comm)
            if self.timing:
                self.timings['overlapped'].append((t0, done))
        self._inflight[id(p)] = (work, done, op)

    def _grad_arrived(self, p):
        if self._world() > 1 and p.grad is not None:
            self._reduce_big(p, grad_ready_event(p.grad))

    def average(self, group=None):
        if group is not None:
            self.group = group
        w = self._world()
        # large parameters: the ones whose hook did not fire (no gradient on this rank) go first, as zeros
        if w > 1:
            for p in self.big:
                if id(p) not in self._inflight:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                    self._reduce_big(p)
        src, dst = [], []
        for p, v in zip(self.small, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        if w > 1 and self.flat is not None:
            op = self._avg_op()
            t0 = None
            if self.timing and self.flat.is_cuda:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
            dist.all_reduce(self.flat, op=op, group=self.group)
            if op != dist.ReduceOp.AVG:
                self.flat.div_(w)
            if t0 is not None:
                t1.record()
                self.timings['flat'].append((t0, t1))
        for p, v in zip(self.small, self.views):
            p.grad = v
        for p in self.big:
            work, done, op = self._inflight.pop(id(p), (None, None, None))
            if work is None:
                continue
            if done is not None:
                torch.cuda.current_stream(p.grad.device).wait_event(done)
            else:
                work.wait()
            if op != dist.ReduceOp.AVG:
                p.grad.div_(w)
        _GRAD_READY.clear()            # marks that no hook consumed belong to this pass: never to a later tensor
        return self.flat


_AVERAGERS = {}


def average_gradients(params):
    """In-place mean of .grad over all ranks (a GradientAverager cached per parameter list)."""
    params = list(params)
    if world() == 1:
        return
    key = tuple(id(p) for p in params)
    if key not in _AVERAGERS:
        for old in _AVERAGERS.values():
            old.close()
        _AVERAGERS.clear()
        # called after the backward pass: nothing to overlap with, one flat message
        _AVERAGERS[key] = GradientAverager(params, overlap=False)
    _AVERAGERS[key].average()


def all_gather_rows(t, sizes=None):
    """Concatenate per-rank row blocks (rendered pixels of an image sharded by rays), ragged allowed.
    ``sizes``: the row count of every rank when the caller can derive it (a chunk deal or ``shard_slice`` of a list
    every rank knows) -- then ONE collective and no host read; without it the counts are exchanged first."""
    w = world()
    if w == 1:
        return t
    if sizes is None:
        n = torch.tensor([t.shape[0]], device=t.device)
        got = [torch.zeros_like(n) for _ in range(w)]
        dist.all_gather(got, n)
        sizes = torch.cat(got).tolist()                  # the one host read of this variant
    sizes = [int(s) for s in sizes]
    if len(sizes) != w or sizes[rank()] != t.shape[0]:
        raise ValueError('all_gather_rows: sizes %r do not describe this rank\'s %d rows' % (sizes, t.shape[0]))
    m = max(sizes)
    if m == 0:
        return t
    if t.shape[0] == m:
        pad = t.contiguous()
    else:
        pad = torch.zeros((m,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        pad[:t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(out, pad)
    if all(s == m for s in sizes):
        return torch.cat(out, 0)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], 0)
