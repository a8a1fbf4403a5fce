"""Data-parallel gradient averaging for one process per GPU (RCCL over xGMI).

Replaces the reference's `nn.parallel.DistributedDataParallel(..., find_unused_parameters=True)`
(tools/runner.py:83-93).  Design for MI355X's point-to-point xGMI fabric:
  * gradients accumulate IN PLACE into a few large contiguous fp32 buckets (parameters' `.grad`
    are views into them), so a collective moves one large message per bucket instead of many
    25 MB ones, and no flatten/unflatten copies exist;
  * buckets are filled in reverse parameter order (the order backward produces gradients); a
    bucket's all-reduce is launched asynchronously from the autograd hook of its last gradient,
    overlapping with the rest of backward;
  * parameters that never receive a gradient (AdaPoinTr: `base_model.encoder.norm.*`,
    `base_model.query_ranking.*`) are discovered on the first step and left out of
    the buckets — there is no per-iteration unused-parameter bitmap all-reduce; their `.grad`
    stays None so the optimizer skips them exactly as it does under the reference's DDP;
  * `finalize()` (called by the runner after backward) waits for the collectives, launches any
    bucket whose hooks did not all fire (rare control-flow change) and divides by world size.
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class FlatDDP(nn.Module):
    def __init__(self, module, bucket_mb=64, process_group=None):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_bytes = int(bucket_mb * 1024 * 1024)
        self._buckets = None       # list of dict(flat, params, pending, handle)
        self._param_bucket = {}
        self._hooks = []
        self._discovered = False
        # identical start: rank 0's parameters and buffers everywhere (as DDP does at wrap time)
        if self.world_size > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    # ------------------------------------------------------------------ bucket construction
    def _build_buckets(self, used_params):
        buckets, cur, cur_bytes = [], [], 0
        for p in reversed(used_params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= self.bucket_bytes:
                buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            buckets.append(cur)
        self._buckets = []
        self._param_bucket = {}
        for plist in buckets:
            n = sum(p.numel() for p in plist)
            flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
            b = dict(flat=flat, params=plist, pending=len(plist), handle=None, views=[])
            off = 0
            for p in plist:
                v = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v
                b["views"].append(v)
                off += p.numel()
                self._param_bucket[p] = b
            self._buckets.append(b)
        for p in self._param_bucket:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        b = self._param_bucket.get(p)
        if b is None:
            return
        b["pending"] -= 1
        if b["pending"] == 0 and self.world_size > 1:
            b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.process_group,
                                          async_op=True)

    # ------------------------------------------------------------------------- step protocol
    def zero_grad(self, set_to_none=True):
        """Zero the buckets and re-attach the `.grad` views; unused parameters get grad None."""
        if self._buckets is None:
            self.module.zero_grad(set_to_none=True)
            return
        for b in self._buckets:
            b["flat"].zero_()
            b["pending"] = len(b["params"])
            b["handle"] = None
            for p, v in zip(b["params"], b["views"]):
                p.grad = v
        for p in self.module.parameters():
            if p not in self._param_bucket:
                p.grad = None

    def finalize(self):
        """Call once after backward: completes the gradient averaging for this step."""
        if not self._discovered:
            # First step: learn which parameters take part, average them without overlap.
            used = [p for p in self.module.parameters() if p.requires_grad and p.grad is not None]
            self._build_buckets(used)
            self._discovered = True
            if self.world_size > 1:
                for b in self._buckets:
                    dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.process_group)
                    b["flat"].div_(self.world_size)
            return
        if self.world_size > 1:
            for b in self._buckets:
                if b["handle"] is None:  # some hook did not fire: reduce what is there (zeros included)
                    b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.process_group,
                                                  async_op=True)
            for b in self._buckets:
                b["handle"].wait()
                b["flat"].div_(self.world_size)
            # a parameter outside the buckets that suddenly received a gradient
            for p in self.module.parameters():
                if p.requires_grad and p.grad is not None and p not in self._param_bucket:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.process_group)
                    p.grad.div_(self.world_size)
        # parameters inside a bucket whose hook did not fire keep a zero gradient (DDP semantics)

    def grad_buffers(self):
        return [b["flat"] for b in (self._buckets or [])]
