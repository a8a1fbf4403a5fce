"""AdamW + global gradient-norm clipping for all parameters in three kernel launches (csrc/optimizer.hip).

`ClipAdamW` IS a `torch.optim.AdamW` (same constructor, param groups, state keys `step` / `exp_avg` / `exp_avg_sq`, so
checkpoints and LR schedulers are unaffected); it adds `step_clipped(max_norm)`, which does what the reference's
`clip_grad_norm_(params, max_norm, 2)` followed by `optimizer.step()` does (tools/runner.py:208-212) — except that the
gradients themselves are left unscaled (the coefficient is applied on the fly; they are dropped right afterwards).
Parameters that are not fp32 CUDA tensors with dense contiguous gradients send the whole step down the stock path."""
import numpy as np
import torch

from .. import _native

_ROW = 64      # bytes per tensor row of the device table (ptops_adamw_row_bytes)


class ClipAdamW(torch.optim.AdamW):
    def __init__(self, params, **kwargs):
        self._plan = self._eager_plan = None    # before super().__init__: it calls add_param_group
        self._ring = 0
        params = list(params)
        flat = [p for g in params for p in g["params"]] if params and isinstance(params[0], dict) else params
        # the stock `step()` (PTOPS_FUSED_ADAMW=0, an external caller, a parameter set `step_clipped` does not support) stays
        # torch's fused multi-tensor AdamW where that exists, not the slower for-each form
        if "fused" not in kwargs and flat and all(isinstance(p, torch.Tensor) and p.is_cuda and p.is_floating_point() for p in flat):
            kwargs["fused"] = True
        super().__init__(params, **kwargs)

    def _sync_steps(self):
        """Write the plan's step counts back into state[p]["step"] (torch's format) before anyone else reads them."""
        if getattr(self, "_plan", None) is not None:
            steps = self._plan["steps"]
            tensors = [self.state[p]["step"] for p in self._plan["params"]]
            if len(tensors) > 8 and float(steps.min()) == float(steps.max()) and all(t.is_cuda for t in tensors):
                # the usual case (every parameter stepped every time): two multi-tensor launches instead of one per tensor
                torch._foreach_zero_(tensors)
                torch._foreach_add_(tensors, float(steps[0]))
            else:
                for t, n in zip(tensors, steps):
                    t.fill_(float(n))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def step(self, closure=None):
        self._sync_steps()
        self._plan = None                              # the stock step advances the counters itself
        return super().step(closure)

    # the plan caches the addresses of the moment tensors: anything that may replace them drops it
    def load_state_dict(self, state_dict):
        self._plan = self._eager_plan = None
        return super().load_state_dict(state_dict)

    def add_param_group(self, param_group):
        self._sync_steps()
        self._plan = None
        return super().add_param_group(param_group)

    # ------------------------------------------------------------------ stock path (CPU tensors, exotic gradients)
    def _stock(self, max_norm):
        params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        norm = torch.nn.utils.clip_grad_norm_(params, max_norm, norm_type=2) if max_norm and max_norm > 0 and params else None
        super().step()
        return norm

    def _supported(self, params, check_grads=True):
        if not params or _native.backend_for_testing() is not None:
            return False
        dev = params[0].device
        if not self.uniform_hyperparameters():
            return False
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable"):
                return False
        return all(p.is_cuda and p.device == dev and p.dtype == torch.float32 and p.is_contiguous()
                   and (not check_grads or (p.grad.dtype == torch.float32 and not p.grad.is_sparse
                                            and p.grad.is_contiguous() and p.grad.device == dev)) for p in params)

    # ------------------------------------------------------------------ the plan: everything that does not change per step
    def _make_plan(self, params, group_of):
        dev = params[0].device
        n = len(params)
        chunk = int(_native.load().ptops_adamw_chunk_elements())
        assert int(_native.load().ptops_adamw_row_bytes()) == _ROW
        group = {id(p): g for g in self.param_groups for p in g["params"]}
        for p in params:
            st = self.state[p]
            if len(st) == 0:
                # where torch's own `_init_group` puts it: the fused / capturable multi-tensor AdamW reads `step` through a
                # device pointer, so a later stock `step()` (external caller, unsupported parameter set) must find it there
                g = group[id(p)]
                on_device = bool(g.get("fused") or g.get("capturable"))
                st["step"] = (torch.zeros((), dtype=torch.float32, device=p.device) if on_device
                              else torch.tensor(0.0, dtype=torch.float32))
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        rows = np.zeros((n, _ROW // 8), dtype=np.uint64)                 # 8 x 8 bytes per row
        rows[:, 0] = [p.data_ptr() for p in params]
        rows[:, 2] = [self.state[p]["exp_avg"].data_ptr() for p in params]
        rows[:, 3] = [self.state[p]["exp_avg_sq"].data_ptr() for p in params]
        numel = np.array([p.numel() for p in params], dtype=np.int64)
        rows[:, 4] = numel.view(np.uint64)
        counts = (numel + chunk - 1) // chunk
        tensor_of = np.repeat(np.arange(n, dtype=np.int32), counts)
        first = np.cumsum(counts) - counts
        offset = (np.arange(int(counts.sum()), dtype=np.int64) - np.repeat(first, counts)) * chunk
        chunks = np.zeros((len(tensor_of), 2), dtype=np.int64)
        chunks[:, 0] = tensor_of.astype(np.int64)                        # int32 tensor index + int32 zero padding
        chunks[:, 1] = offset
        plan = {
            "key": tuple(id(p) for p in params), "params": params, "group_of": np.asarray(group_of), "rows": rows,
            # step counts live here between checkpoints (335 CPU-tensor increments per step cost a millisecond of host
            # time); `_sync_steps` writes them back into state[p]["step"] whenever somebody else may look
            "steps": torch.stack([self.state[p]["step"].detach().reshape(()).to("cpu", torch.float64)
                                  for p in params]).numpy().copy(),
            "n_chunks": len(tensor_of),
            "chunks_dev": torch.from_numpy(chunks.view(np.uint8).reshape(-1)).to(dev),
            "rows_dev": torch.empty(n * _ROW, dtype=torch.uint8, device=dev),
            # pinned staging buffers in a ring: an async copy may still be reading the one of the previous steps
            "rows_host": [torch.empty(n * _ROW, dtype=torch.uint8).pin_memory() for _ in range(4)],
            "rows_event": [torch.cuda.Event() for _ in range(4)],
            "rows_used": [False] * 4,
            "partial": torch.empty(len(tensor_of), dtype=torch.float32, device=dev),
            "norm_coef": torch.empty(2, dtype=torch.float32, device=dev),
        }
        return plan

    def _with_grad(self):
        params, group_of = [], []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is not None:
                    params.append(p)
                    group_of.append(gi)
        return params, group_of

    def _activate(self, plan):
        """Make `plan` the current one; the step counters travel through state[p]["step"] (torch's format)."""
        if self._plan is not plan:
            self._sync_steps()
            plan["steps"] = torch.stack([self.state[p]["step"].detach().reshape(()).to("cpu", torch.float64)
                                         for p in plan["params"]]).numpy().copy()
            self._plan = plan

    def _fill_and_upload(self, plan):
        """Advance the step counters, write this step's hyper-parameters into the table and send it to the device
        (asynchronously, on the current stream, from a ring of pinned buffers)."""
        rows = plan["rows"]
        plan["steps"] += 1.0
        steps = plan["steps"]
        beta1, beta2 = self.param_groups[0]["betas"]
        lr = np.array([g["lr"] for g in self.param_groups], dtype=np.float32)[plan["group_of"]]
        wd = np.array([g["weight_decay"] for g in self.param_groups], dtype=np.float32)[plan["group_of"]]
        f32 = rows.view(np.float32).reshape(len(plan["params"]), _ROW // 4)
        f32[:, 10] = lr
        f32[:, 11] = wd
        f32[:, 12] = (1.0 - beta1 ** steps).astype(np.float32)
        f32[:, 13] = np.sqrt(1.0 - beta2 ** steps).astype(np.float32)
        slot = self._ring
        self._ring = (self._ring + 1) % len(plan["rows_host"])
        host = plan["rows_host"][slot]
        if plan["rows_used"][slot]:
            plan["rows_event"][slot].synchronize()      # the copy that read this buffer four steps ago (long done)
        host.numpy()[:] = rows.view(np.uint8).reshape(-1)
        plan["rows_dev"].copy_(host, non_blocking=True)
        plan["rows_event"][slot].record()
        plan["rows_used"][slot] = True

    def _launch(self, plan, max_norm):
        beta1, beta2 = self.param_groups[0]["betas"]
        _native.adamw_clip_step(plan["rows_dev"], len(plan["params"]), plan["chunks_dev"], plan["n_chunks"],
                                float(max_norm) if max_norm else 0.0, beta1, beta2, self.param_groups[0]["eps"],
                                plan["partial"], plan["norm_coef"])

    def _pre_hooks(self):
        # the hooks `Optimizer.step` would run (register_step_pre_hook / register_step_post_hook, global ones included)
        from torch.optim.optimizer import _global_optimizer_pre_hooks
        for hook in list(_global_optimizer_pre_hooks.values()) + list(self._optimizer_step_pre_hooks.values()):
            hook(self, (), {})

    def _post_hooks(self):
        from torch.optim.optimizer import _global_optimizer_post_hooks
        self._opt_called = True                        # what LRScheduler's wrapper around step() records
        for hook in list(self._optimizer_step_post_hooks.values()) + list(_global_optimizer_post_hooks.values()):
            hook(self, (), {})

    @torch.no_grad()
    def step_clipped(self, max_norm):
        """clip_grad_norm_(all parameters with a gradient, max_norm) + step().  -> the gradient norm (0-dim tensor)."""
        params, group_of = self._with_grad()
        key = tuple(id(p) for p in params)
        stale = self._plan is None or self._plan["key"] != key or self._plan.get("graph", False)
        if stale:                                                       # the full eligibility check once per parameter set
            if not self._supported(params):
                self._sync_steps()
                self._plan = None
                return self._stock(max_norm)
        if stale or any(self.state[p]["exp_avg"].data_ptr() != int(a) for p, a in zip(params[:4], self._plan["rows"][:4, 2])):
            kept = getattr(self, "_eager_plan", None)
            if (kept is not None and kept["key"] == key
                    and all(self.state[p]["exp_avg"].data_ptr() == int(a) for p, a in zip(params[:4], kept["rows"][:4, 2]))):
                self._activate(kept)                  # eager and graphed steps alternating: the eager plan is still good
            else:
                self._sync_steps()                    # the old plan's counters go back into the state first
                self._plan = self._eager_plan = self._make_plan(params, group_of)
        plan = self._plan
        self._pre_hooks()
        plan["rows"][:, 1] = [p.grad.data_ptr() for p in params]
        self._fill_and_upload(plan)
        self._launch(plan, max_norm)
        self._post_hooks()
        return plan["norm_coef"][0]

    # ------------------------------------------------------------------ the same step inside a captured HIP graph
    # (utils/graph_runner.py::GraphedTrainStep).  A graph bakes in addresses, not values: the kernels read everything
    # that changes per step (learning rate, weight decay, bias corrections) from the device table, which is rewritten
    # BEFORE each replay by an ordinary stream-ordered copy; the gradients live at fixed addresses of the graph's pool.
    @torch.no_grad()
    def graph_plan(self, params):
        """Before the capture: the plan for `params` (those that will hold a gradient), or None if the fused step does
        not support them."""
        group_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        with_grad = [p for g in self.param_groups for p in g["params"] if any(p is q for q in params)]
        if len(with_grad) != len(params) or not with_grad:
            return None
        if not self._supported(with_grad, check_grads=False):      # the gradients' layout is checked inside the capture
            return None
        self._sync_steps()
        self._plan = None
        plan = self._make_plan(with_grad, [group_of[id(p)] for p in with_grad])
        plan["graph"] = True
        return plan

    @torch.no_grad()
    def graph_capture_step(self, plan, max_norm):
        """Inside the capture, where `step_clipped` would run: notes the (fixed) gradient addresses and records the three
        launches.  -> the gradient norm (a tensor that every replay overwrites)."""
        params, _ = self._with_grad()
        if tuple(id(p) for p in params) != plan["key"]:
            raise RuntimeError("ClipAdamW.graph_capture_step: the parameters holding a gradient differ from the plan's")
        for p in params:
            if not (p.grad.dtype == torch.float32 and p.grad.is_contiguous() and not p.grad.is_sparse):
                raise RuntimeError("ClipAdamW.graph_capture_step: unsupported gradient layout")
        plan["grad_ptrs"] = np.array([p.grad.data_ptr() for p in params], dtype=np.uint64)
        self._launch(plan, max_norm)
        return plan["norm_coef"][0]

    @torch.no_grad()
    def graph_before_replay(self, plan):
        """Before every replay (stream-ordered in front of it): counters, hyper-parameters and table upload."""
        if any(self.state[p]["exp_avg"].data_ptr() != int(a) for p, a in zip(plan["params"][:4], plan["rows"][:4, 2])):
            raise RuntimeError("ClipAdamW: the optimizer state was replaced (load_state_dict) under a captured step")
        self._activate(plan)
        self._pre_hooks()
        plan["rows"][:, 1] = plan["grad_ptrs"]
        self._fill_and_upload(plan)

    def graph_after_replay(self, plan):
        self._post_hooks()

    def uniform_hyperparameters(self):
        """betas / eps must be the same in every group for the fused path (they are kernel arguments)."""
        b, e = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        return all(g["betas"] == b and g["eps"] == e for g in self.param_groups)
