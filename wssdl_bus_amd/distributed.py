"""Image-parallel data parallelism: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-GPU (no collectives anywhere).  The hot path
shards by image with no exchange step (section 8e), so the only collective is one gradient
all-reduce (sum -> mean) per optimiser step.  Gradients are packed into a few large flat
buckets: on MI355X's point-to-point xGMI a ring all-reduce is per-link bound, so few large
messages beat many small ones (ResNet-50 variant: ~130 MB fp32 -> 3 buckets of 64 MiB).
"""
import os

import torch
import torch.distributed as dist


class DistContext(object):
    def __init__(self, backend=None, bucket_bytes=64 << 20):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.bucket_bytes = bucket_bytes
        self.backend = backend
        # any-rank has-grad patterns, per context (ids of freed parameters can be reused by a later model:
        # the cache must not outlive the parameters it was learnt on -- cleared by GradOverlap.remove / shutdown)
        self._any_cache = {}
        # WSSDL_FORCE_DIST=1: run the collective code path even with one rank (RCCL smoke test
        # on a single-GPU box)
        self.enabled = self.world_size > 1 or bool(os.environ.get("WSSDL_FORCE_DIST"))
        if self.enabled and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                # "nccl" IS RCCL on ROCm.  WSSDL_DIST_BACKEND=gloo lets the multi-rank code path
                # be exercised with several ranks on one GPU (RCCL refuses duplicate devices).
                backend = os.environ.get("WSSDL_DIST_BACKEND") or (
                    "nccl" if torch.cuda.is_available() else "gloo")
            self.backend = backend
            if torch.cuda.is_available():
                torch.cuda.set_device(self.local_rank % torch.cuda.device_count())
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)

    # ---- sharding: whole images per rank, no data-path collective ----
    def shard_images(self, n_global):
        """Indices of the images this rank owns out of n_global (contiguous blocks)."""
        per = n_global // self.world_size
        rem = n_global % self.world_size
        start = self.rank * per + min(self.rank, rem)
        return list(range(start, start + per + (1 if self.rank < rem else 0)))

    def seed(self, base):
        """RNG seed per rank = RNG_SEED + rank."""
        return int(base) + self.rank

    # ---- the one collective ----
    def allreduce_gradients(self, params):
        """In-place mean of .grad across ranks, bucketed.  Parameters whose grad is None on
        this rank (e.g. RPN weights in a MIL-only step) contribute zeros, so every rank issues
        the same collectives; a parameter that had no gradient on ANY rank gets its grad set
        back to None afterwards, so the optimiser skips it exactly as on one GPU (and as the
        reference's apply_gradients skips a None gradient, train_bus.py:297-301)."""
        if not self.enabled:
            return
        bucket, size = [], 0
        handles = []
        for p in params:
            bucket.append(p)
            size += p.numel() * p.element_size()
            if size >= self.bucket_bytes:
                handles.append(self._launch(bucket))
                bucket, size = [], 0
        if bucket:
            handles.append(self._launch(bucket))
        for h in handles:
            self._write_back(*h)
        self.poll_patterns()

    _flag_cache = {}

    def _flags(self, had, like):
        """[len(had)] tensor of 1.0 / 0.0 on like's device (cached per pattern: no H2D per step)."""
        key = (tuple(had), str(like.device), like.dtype)
        t = DistContext._flag_cache.get(key)
        if t is None:
            t = torch.tensor([1.0 if h else 0.0 for h in had], dtype=like.dtype, device=like.device)
            DistContext._flag_cache[key] = t
        return t

    def _launch(self, ps):
        """One flat bucket = the gradients followed by one has-grad flag per parameter."""
        had = [p.grad is not None for p in ps]
        parts = [p.grad.reshape(-1) if h else torch.zeros(p.numel(), dtype=p.dtype, device=p.device)
                 for p, h in zip(ps, had)]
        parts.append(self._flags(had, ps[0]))
        flat = torch.cat(parts)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        return flat, work, list(ps), tuple(had)

    # Which parameters of a bucket had a gradient on ANY rank decides `grad = None`, a host decision.
    # All ranks run the same kind of step (combined / supervised / weak) in lock-step, so that pattern
    # is a function of the local has-grad pattern: it is read back ONCE per (bucket, local pattern) --
    # the first step of each kind -- and cached.  Later steps compare the reduced flags with the cached
    # pattern ON THE DEVICE and add the differences to a counter that poll_patterns() looks at without
    # synchronising (the copy of the previous poll, once its event has completed): a rank whose
    # pattern ever departs raises one step later instead of training on.
    def _write_back(self, flat, work, ps, had=None):
        work.wait()
        flags = flat[flat.numel() - len(ps):]
        key = (tuple(id(p) for p in ps), had)
        any_grad = self._any_cache.get(key) if had is not None else None
        if any_grad is None:
            any_grad = tuple(a > 0 for a in flags.cpu().tolist())     # first step of this kind only
            if had is not None:
                self._any_cache[key] = any_grad
        else:
            diff = ((flags > 0).to(flat.dtype) - self._flags(any_grad, flat)).abs().sum()
            if getattr(self, "_mismatch", None) is None:
                self._mismatch = torch.zeros((), dtype=flat.dtype, device=flat.device)
            self._mismatch.add_(diff)
        flat.div_(self.world_size)
        off = 0
        for p, a in zip(ps, any_grad):
            n = p.numel()
            if a:
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
            else:
                p.grad = None
            off += n

    def poll_patterns(self):
        """Sync-free look at the has-grad mismatch counter (see _write_back)."""
        m = getattr(self, "_mismatch", None)
        if m is None:
            return
        ev = getattr(self, "_mismatch_event", None)
        if ev is not None and (not m.is_cuda or ev.query()):
            self._mismatch_event = None
            if float(self._mismatch_host) != 0.0:
                raise RuntimeError("data-parallel has-grad pattern changed between steps of one kind: "
                                   "ranks disagree on which parameters received a gradient")
        if getattr(self, "_mismatch_event", None) is None:
            if m.is_cuda:
                if getattr(self, "_mismatch_host", None) is None:
                    self._mismatch_host = torch.zeros((), dtype=m.dtype).pin_memory()
                self._mismatch_host.copy_(m, non_blocking=True)
                self._mismatch_event = torch.cuda.Event()
                self._mismatch_event.record()
            else:
                self._mismatch_host = m.clone()
                self._mismatch_event = True

    def overlap(self, params):
        """Bucketed all-reduce overlapped with backward: see GradOverlap."""
        return GradOverlap(self, params)

    def barrier(self):
        if self.enabled:
            dist.barrier()

    def max_over_ranks(self, value):
        if not self.enabled:
            return value
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, value):
        """[value of rank 0, ..., value of rank world_size - 1] on every rank (one all-gather)."""
        if not self.enabled:
            return [float(value)]
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        out = [torch.zeros_like(t) for _ in range(self.world_size)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def count_ranks(self):
        """How many ranks the collective backend really connects: every rank adds 1 (an all-reduce over
        RCCL / gloo), so a scaling record shows that N processes took part, not just that N were asked for."""
        return int(round(self.sum_over_ranks(1.0)))

    def sum_over_ranks(self, value):
        if not self.enabled:
            return value
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def shutdown(self):
        self._any_cache.clear()
        if self.enabled and dist.is_initialized():
            dist.destroy_process_group()


class GradOverlap(object):
    """Gradient all-reduce overlapped with the backward pass.

    Parameters are packed (in reverse registration order, i.e. roughly the order in which
    backward produces their gradients) into buckets of ``ctx.bucket_bytes``.  A post-accumulate
    hook on every parameter counts its bucket down; when the last gradient of a bucket has
    been accumulated the bucket is flattened and its all-reduce is launched asynchronously on
    RCCL's stream while backward keeps running.  ``finish()`` (called before the optimiser step)
    launches whatever is left -- buckets holding parameters that received no gradient in this
    backward, e.g. the RPN weights in a MIL-only step, are completed with zeros so that every
    rank issues the same collectives in the same order -- waits, and writes the means back.
    Every bucket carries one has-grad flag per parameter: a parameter without a gradient on
    every rank ends with ``grad = None`` again, so Adam skips it as it does on one GPU."""

    def __init__(self, ctx, params):
        self.ctx = ctx
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= ctx.bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {}
        for b, ps in enumerate(self.buckets):
            for p in ps:
                self.bucket_of[id(p)] = b
        self.handles = []
        if ctx.enabled:
            self.handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        self.timing, self._spans, self.finishes, self.late_buckets = False, [], 0, 0
        self._reset()

    def _reset(self):
        self.pending = [len(ps) for ps in self.buckets]
        self.launched = [None] * len(self.buckets)
        self.next_to_launch = 0

    def _hook(self, p):
        b = self.bucket_of[id(p)]
        self.pending[b] -= 1
        # launch strictly in bucket order so that all ranks issue identical collectives
        while (self.next_to_launch < len(self.buckets) and self.pending[self.next_to_launch] == 0):
            self._launch(self.next_to_launch)
            self.next_to_launch += 1

    def _launch(self, b):
        flat, work, _, had = self.ctx._launch(self.buckets[b])
        self.launched[b] = (flat, work, had)

    def finish(self):
        """Call after backward, before optimizer.step()."""
        if not self.ctx.enabled:
            return
        self.late_buckets += len(self.buckets) - self.next_to_launch
        for b in range(self.next_to_launch, len(self.buckets)):
            self._launch(b)
        # exposed all-reduce time = how long the compute stream stands still here waiting for the collectives that
        # backward did not hide (RCCL: work.wait() makes the current stream wait, events on that stream bracket it; gloo:
        # the wait blocks the host, a host clock brackets it).  Collected only while `timing` is on (bench.py).
        t0 = self._mark() if self.timing else None
        for b in range(len(self.buckets)):
            self.launched[b][1].wait()
        if self.timing:
            self._spans.append((t0, self._mark()))
        for b, ps in enumerate(self.buckets):
            flat, work, had = self.launched[b]
            self.ctx._write_back(flat, work, ps, had)
        self.ctx.poll_patterns()
        self.finishes += 1
        self._reset()

    def _mark(self):
        p = self.params[0] if self.params else None
        if p is not None and p.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def start_timing(self):
        self.timing, self._spans, self.finishes, self.late_buckets = True, [], 0, 0

    def stats(self):
        """After a synchronize: what one scaling record needs to explain itself -- buckets and their bytes (gradients
        + has-grad flags, as sent), all-reduce calls per finish(), exposed wait per finish() in ms, and how many buckets
        were still unlaunched when backward ended (launched by finish(): never overlapped)."""
        spans = []
        for a, b in self._spans:
            spans.append(a.elapsed_time(b) if hasattr(a, "elapsed_time") else (b - a) * 1e3)
        sizes = [sum(p.numel() * p.element_size() for p in ps) + len(ps) * ps[0].element_size() for ps in self.buckets]
        n = max(self.finishes, 1)
        return dict(buckets=len(self.buckets), bucket_bytes=sizes, allreduce_bytes_per_finish=int(sum(sizes)),
                    finishes=self.finishes, exposed_wait_ms_per_finish=(sum(spans) / len(spans) if spans else 0.0),
                    exposed_wait_ms_max=(max(spans) if spans else 0.0),
                    late_buckets_per_finish=self.late_buckets / n, bucket_bytes_cap=int(self.ctx.bucket_bytes))

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []
        self.ctx._any_cache.clear()       # learnt on this model's parameters
