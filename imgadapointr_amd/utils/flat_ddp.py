"""Data-parallel gradient averaging for one process per GPU (RCCL over xGMI).

Replaces the reference's `nn.parallel.DistributedDataParallel(..., find_unused_parameters=True)`
(tools/runner.py:83-93).  Design for MI355X's point-to-point xGMI fabric:
  * gradients are averaged in a few large contiguous fp32 buckets, so a collective moves one large
    message per bucket instead of many 25 MB ones; the buckets shrink towards the end of backward (64, 33, 16, 8, 8 MB
    for AdaPoinTr), because the last one to complete is the only collective that cannot hide behind backward;
  * buckets are laid out in reverse parameter order (the order backward produces gradients).  During
    backward every parameter's `.grad` is the tensor autograd produced (no accumulate-into-zero kernels:
    `.grad` is None when backward starts); when the LAST gradient of a bucket arrives, its hook packs
    the bucket with one multi-tensor copy, launches the bucket's all-reduce asynchronously (it overlaps
    the rest of backward) and re-points the `.grad`s at views of the bucket, so nothing is unpacked;
  * parameters that never receive a gradient (AdaPoinTr: `base_model.encoder.norm.*`,
    `base_model.query_ranking.*`) are discovered on the first step and left out of
    the buckets — there is no per-iteration unused-parameter bitmap all-reduce; their `.grad`
    stays None so the optimizer skips them exactly as it does under the reference's DDP;
  * `finalize()` (called by the runner after backward) waits for the collectives, launches any
    bucket whose hooks did not all fire (rare control-flow change; missing gradients count as zero)
    and divides by world size;
  * `set_sync(False)` turns the averaging off for a backward (gradient accumulation over several
    micro-batches: only the last one is averaged, as with DDP's `no_sync`);
  * the participating set is agreed on by ALL ranks (one bitmap all-reduce on the first step), so every rank
    builds the same buckets even if a data-dependent branch skipped a parameter on one of them;
  * a parameter outside the buckets that receives a gradient later is never averaged by an extra collective
    (a collective only some ranks enter hangs the job): its gradient is dropped on that rank, so the parameter
    stays identical everywhere, and `check_late_gradients()` — a collective the runner calls at its log lines —
    raises the same error on every rank; `rediscover()` re-learns the set on the next step;
  * buffers (BatchNorm running statistics) are broadcast from rank 0 at wrap time and by `sync_buffers()`, which
    the runner calls before validation: DDP's `broadcast_buffers=True` makes rank 0's statistics the ones every
    rank evaluates with; here they are synchronised where they are read instead of once per forward.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _native


class FlatDDP(nn.Module):
    def __init__(self, module, bucket_mb=None, process_group=None, single_rank_collectives=False):
        """`single_rank_collectives`: issue the broadcasts / all-reduces even in a world of one rank (a way to run the
        exact RCCL call sequence on a one-GPU box, tests/test_runner_gpu.py); off, a single rank skips them."""
        super().__init__()
        if bucket_mb is None:                     # PTOPS_DDP_BUCKET_MB: A/B of the bucket schedule (default 64)
            import os
            bucket_mb = float(os.environ.get("PTOPS_DDP_BUCKET_MB", "64"))
        self.module = module
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._collective = self.world_size > 1 or (single_rank_collectives and dist.is_initialized())
        self.bucket_bytes = int(bucket_mb * 1024 * 1024)
        self._buckets = None       # list of dict(flat, params, views, pending, handle, packed)
        self._param_bucket = {}
        self._hooks = []
        self._discovered = False
        self._sync = True
        self._late = set()
        self._every_param = list(module.parameters())
        self._learning, self._relearn = True, False
        self._allow_slim = True
        self._outside = None          # trainable parameters outside the buckets (name, parameter), cached by finalize()
        # RCCL averages inside the collective; gloo (CPU tests) only sums, so divide afterwards there
        self._avg_in_collective = (self._collective and dist.get_backend(process_group) == "nccl")
        # identical start: rank 0's parameters and buffers everywhere (as DDP does at wrap time)
        if self._collective:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def set_sync(self, sync):
        """Called before every backward: `sync` False for all but the last micro-batch of an accumulated update.

        Gradient accumulation (any backward that is not averaged) rules out the one-hook-per-bucket mode for good: there
        every `.grad` is non-None from the first micro-batch on, so the slim hook's completeness check would pass before
        the bucket's remaining gradients of THIS backward have accumulated; the per-parameter counting hooks stay."""
        self._sync = bool(sync)
        if not self._sync and self._allow_slim:
            self._allow_slim = False
            if self._buckets and not self._learning:
                for b in self._buckets:
                    b["last"] = None
                self._set_hooks(slim=False)

    # ------------------------------------------------------------------ bucket construction
    def _build_buckets(self, used_params):
        # Backward order; bucket sizes shrink towards the end of backward: the bucket that completes last (the first
        # layers' gradients) is the one collective nothing can overlap, so it should be small, while the early ones
        # have the rest of backward to hide behind and can be large: cap = min(bucket_bytes, half of what is left),
        # at least min(8 MB, bucket_bytes).  AdaPoinTr (130 MB): 64, 33, 16, 8, 8 MB.
        sizes = [p.numel() * p.element_size() for p in reversed(used_params)]
        remaining = sum(sizes)
        floor_bytes = min(8 << 20, self.bucket_bytes)
        cap = max(floor_bytes, min(self.bucket_bytes, remaining // 2))
        buckets, cur, cur_bytes = [], [], 0
        for p, nbytes in zip(reversed(used_params), sizes):
            cur.append(p)
            cur_bytes += nbytes
            if cur_bytes >= cap:
                buckets.append(cur)
                remaining -= cur_bytes
                cur, cur_bytes = [], 0
                cap = max(floor_bytes, min(self.bucket_bytes, remaining // 2))
        if cur:
            buckets.append(cur)
        self._buckets = []
        self._param_bucket = {}
        for plist in buckets:
            n = sum(p.numel() for p in plist)
            flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
            b = dict(flat=flat, params=plist, pending=len(plist), handle=None, views=[], packed=False, last=None)
            off = 0
            for p in plist:
                b["views"].append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                self._param_bucket[p] = b
            self._buckets.append(b)
        # Hooks: the first averaged backward after this runs one hook per parameter and LEARNS which parameter completes
        # each bucket (a static graph hands its gradients over in the same order every step); from then on only those
        # parameters carry a hook (5 Python callbacks per backward instead of 330: the autograd thread enqueues the end
        # of backward noticeably faster).  The slim hook checks that its bucket really is complete; if the order ever
        # changes the bucket is finished by finalize() without overlap and the order is learned again.
        self._learning = True
        for p in self._param_bucket:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _pack(self, b):
        """Gather the bucket's gradients into its flat buffer (one multi-tensor copy) and make the
        `.grad`s views of it; a parameter without a gradient contributes zeros."""
        _native.fold_flush()        # bias / LayerNorm gradients may still be queued folds (tools/runner.py)
        src, dst = [], []
        for p, v in zip(b["params"], b["views"]):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                src.append(g)
                dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)
        b["packed"] = True

    def _launch(self, b, async_op=True):
        self._pack(b)
        if self._collective:
            op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
            b["handle"] = dist.all_reduce(b["flat"], op=op, group=self.process_group, async_op=async_op)

    def _on_grad(self, p):
        """Learning mode: one call per parameter and backward."""
        if not self._sync:
            return
        b = self._param_bucket.get(p)
        if b is None:
            return
        b["pending"] -= 1
        if b["pending"] == 0:
            b["last"] = p
            self._launch(b)

    def _on_last(self, p):
        """Slim mode: called for the parameter that completed its bucket when the order was learned."""
        if not self._sync:
            return
        b = self._param_bucket.get(p)
        if b is None or b["packed"]:
            return
        if all(q.grad is not None for q in b["params"]):
            self._launch(b)
        else:                                   # the order changed: finalize() completes this bucket; learn again
            self._relearn = True

    def _set_hooks(self, slim):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if slim:
            for b in self._buckets:
                self._hooks.append(b["last"].register_post_accumulate_grad_hook(self._on_last))
        else:
            for p in self._param_bucket:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._learning = not slim

    # ------------------------------------------------------------------------- step protocol
    def zero_grad(self, set_to_none=True):
        """Drop all gradients (`.grad = None`: backward then hands its tensors over without an add)."""
        for p in self._every_param:                  # cached: walking the module tree costs 0.7 ms of host time per step
            p.grad = None
        for b in (self._buckets or []):
            b["pending"] = len(b["params"])
            b["handle"] = None
            b["packed"] = False

    def _agree_on_used(self, trainable):
        """First step: the union over ranks of the parameters that received a gradient."""
        used = [p.grad is not None for p in trainable]
        if self._collective and trainable:
            dev = trainable[0].device
            bitmap = torch.tensor(used, dtype=torch.int32).to(dev)
            dist.all_reduce(bitmap, op=dist.ReduceOp.MAX, group=self.process_group)
            used = [bool(v) for v in bitmap.tolist()]
        return [p for p, u in zip(trainable, used) if u]

    def finalize(self):
        """Call once after the (last) backward of a step: completes the gradient averaging."""
        if not self._discovered:
            # First step: learn which parameters take part (rank-consistently), average them without overlap.
            trainable = [p for p in self.module.parameters() if p.requires_grad]
            self._build_buckets(self._agree_on_used(trainable))
            self._discovered = True
            for b in self._buckets:
                self._launch(b, async_op=False)
                if self._collective and not self._avg_in_collective:
                    b["flat"].div_(self.world_size)
            return
        for b in self._buckets:
            if not b["packed"]:          # some hook did not fire, or averaging was off during backward
                self._launch(b)
        if self._sync and self._buckets:
            if self._learning and self._allow_slim and all(b["last"] is not None for b in self._buckets):
                self._set_hooks(slim=True)                  # the order is known: one hook per bucket from now on
            elif not self._learning and self._relearn:
                for b in self._buckets:
                    b["last"] = None
                self._set_hooks(slim=False)
            self._relearn = False
        if self._collective:
            for b in self._buckets:
                if b["handle"] is not None:
                    b["handle"].wait()
                if not self._avg_in_collective:
                    b["flat"].div_(self.world_size)
            # A parameter outside the buckets that suddenly received a gradient: no collective here (other ranks may
            # not have it and would never enter one).  Dropping the local gradient keeps the parameter identical on
            # every rank; check_late_gradients() reports it on all of them.
            if self._outside is None:       # cached: walking the module tree costs 0.2 ms of host time per step
                self._outside = [(n, p) for n, p in self.module.named_parameters()
                                 if p.requires_grad and p not in self._param_bucket]
            for n, p in self._outside:
                if p.grad is not None:
                    p.grad = None
                    self._late.add(n)

    def check_late_gradients(self):
        """Collective (every rank must call it at the same point, e.g. at a log line).  Raises on EVERY rank if any
        rank met a gradient for a parameter that the first step left out of the buckets."""
        flag = torch.tensor([1 if self._late else 0], dtype=torch.int32)
        if self._collective:
            flag = flag.to(next(self.module.parameters()).device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.process_group)
        if int(flag.item()):
            names = sorted(self._late)
            raise RuntimeError(
                "FlatDDP: parameters outside the gradient buckets received a gradient after the first step "
                f"(on this rank: {names if names else 'none, another rank did'}); their updates were skipped. "
                "Call rediscover() to re-learn the participating set on the next step.")

    def rediscover(self):
        """Forget the buckets: the next finalize() learns the participating parameters again (on every rank)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._buckets, self._param_bucket = None, {}
        self._outside = None
        self._learning, self._relearn = True, False
        self._discovered = False
        self._late = set()
        self.zero_grad()

    def sync_buffers(self):
        """Rank 0's buffers (BatchNorm running statistics, counters) everywhere.  Collective."""
        if self._collective:
            for t in self.module.buffers():
                dist.broadcast(t.data, src=0, group=self.process_group)

    def unused_parameter_names(self):
        """Names of the trainable parameters that received no gradient on the first step."""
        return [n for n, p in self.module.named_parameters() if p.requires_grad and p not in self._param_bucket]

    def grad_buffers(self):
        return [b["flat"] for b in (self._buckets or [])]
