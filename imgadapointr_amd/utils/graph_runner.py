"""Validation / inference forward as one HIP graph per input shape.

The reference validates and tests with batch size 1 (tools/runner.py:401-450, 480-530): ~600 kernel launches per cloud
with microseconds of work each, so the forward is bound by launch latency, not by the GPU (7.7 ms eager vs 2.8 ms
replayed on an MI355X at bs = 1; 7.8 vs 4.6 ms at bs = 8).  `GraphedForward` captures `model(*inputs)` once per input
signature under `torch.no_grad()` in eval mode and replays it; the library's ctypes launches go to the capturing stream
like any other kernel.

Safety rules (a graph that silently replays the wrong thing is worse than no graph):
  * capture and replay run under `torch.cuda.device(<device of the inputs>)`, so a model on a GPU that is not the
    current device is captured on ITS stream (otherwise the kernels would run eagerly there while an empty graph is
    recorded on the current device and every later replay would return the first call's outputs);
  * all inputs must live on one device, which is part of the cache key;
  * after capture the graph is put through a self-test (unrelated work, a host synchronisation, a replay on
    perturbed inputs, a replay on the original ones) against eager forwards; a mismatch — an empty graph, a stateful
    forward, a graph node that does not survive replays (MEMSET nodes on this stack) — disables graphing for this
    model and says why;
  * only capture failures (RuntimeError / TypeError from the stream capture) fall back to eager, and the reason is
    logged; an out-of-memory error is re-raised;
  * at most `max_graphs` captured signatures are kept (least recently used is dropped together with its memory pool).
"""
import logging
from collections import OrderedDict

import torch

_log = logging.getLogger(__name__)


def _flatten(out):
    if isinstance(out, torch.Tensor):
        return [out], lambda ts: ts[0]
    if isinstance(out, (tuple, list)) and all(isinstance(t, torch.Tensor) for t in out):
        kind = type(out)
        return list(out), lambda ts: kind(ts)
    return None, None


def input_device(inputs):
    """The one CUDA device all `inputs` live on, or None (CPU tensors, non-tensors, mixed devices)."""
    dev = None
    for t in inputs:
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            return None
        if dev is None:
            dev = t.device
        elif t.device != dev:
            return None
    return dev


def signature(inputs):
    """Cache key of a call: shapes, dtypes and the device of the inputs."""
    return tuple((tuple(t.shape), t.dtype, t.device.type, t.device.index) for t in inputs)


class GraphedForward:
    def __init__(self, model, warmup=2, enabled=True, max_graphs=4, verify=True, rtol=1e-4, atol=1e-5):
        self.model = model
        self.warmup = max(1, warmup)
        self.enabled = enabled
        self.max_graphs = max(1, max_graphs)
        self.verify = verify
        self.rtol, self.atol = rtol, atol
        self._cache = OrderedDict()
        self.replays = 0
        self.disabled_reason = None

    def _eligible(self, inputs):
        return (self.enabled and not torch.is_grad_enabled() and not self.model.training and len(inputs) > 0
                and input_device(inputs) is not None)

    def _disable(self, reason):
        self.enabled = False
        self.disabled_reason = reason
        self._cache.clear()
        _log.warning("GraphedForward: falling back to eager forwards (%s)", reason)

    def _capture(self, inputs, device):
        """Runs on `device` (the caller holds the device guard).  -> cache entry or None (graphing disabled)."""
        static_in = [t.clone() for t in inputs]
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    eager = self.model(*static_in)
            torch.cuda.current_stream(device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.model(*static_in)
        except torch.cuda.OutOfMemoryError:
            raise
        except (RuntimeError, TypeError) as exc:      # a host sync / unsupported call inside the forward
            torch.cuda.synchronize(device)
            self._disable(f"capture failed: {type(exc).__name__}: {exc}")
            return None
        flat, rebuild = _flatten(out)
        if flat is None:
            self._disable("the forward must return a tensor or a flat tuple / list of tensors")
            return None
        if any(t.device != device for t in flat):
            self._disable(f"outputs are not on the inputs' device {device}")
            return None
        if self.verify:
            reason = self._self_test(static_in, graph, flat, eager, device)
            if reason is not None:
                self._disable(reason)
                return None
        return static_in, graph, flat, rebuild

    def _matches(self, flat, eager_out):
        eager_flat, _ = _flatten(eager_out)
        return eager_flat is not None and len(eager_flat) == len(flat) and all(
            (torch.allclose(a, b, rtol=self.rtol, atol=self.atol, equal_nan=True) if a.is_floating_point()
             else torch.equal(a, b)) for a, b in zip(flat, eager_flat))

    def _self_test(self, static_in, graph, flat, eager, device):
        """Replays the fresh graph the way validation will use it and compares with eager forwards.  -> reason or None.

        The sequence is the one that exposes a graph which does not survive real use on this stack: unrelated
        allocations and kernels, a HOST SYNCHRONISATION, then a replay on inputs that differ from the captured ones.
        (A MEMSET node in a replayed HIP graph stops taking effect after a host synchronisation — csrc/core.hip,
        benchmarks/graph_diag3.py: stale outputs from ATen's strided reductions, whose semaphores are zeroed by
        memset.  An empty graph, i.e. kernels that ran eagerly on another device during capture, fails here too.)"""
        originals = [t.clone() for t in static_in]
        scratch = torch.zeros(1024, device=device) + 1.0
        torch.cuda.synchronize(device)
        perturbed = [t.roll(1, dims=max(t.dim() - 2, 0)) if t.numel() > 1 else t.clone() for t in originals]
        want = self.model(*perturbed)
        for s, t in zip(static_in, perturbed):
            s.copy_(t)
        graph.replay()
        torch.cuda.synchronize(device)
        if not self._matches(flat, want):
            return "a replay after a host synchronisation did not reproduce the eager forward on new inputs"
        for s, t in zip(static_in, originals):
            s.copy_(t)
        graph.replay()
        torch.cuda.synchronize(device)
        del scratch
        if not self._matches(flat, eager):
            return "a replay did not reproduce the eager forward"
        return None

    def __call__(self, *inputs):
        if not self._eligible(inputs):
            return self.model(*inputs)
        device = input_device(inputs)
        key = signature(inputs)
        with torch.cuda.device(device):
            entry = self._cache.get(key)
            if entry is None:
                entry = self._capture(inputs, device)
                if entry is None:
                    return self.model(*inputs)
                self._cache[key] = entry
                while len(self._cache) > self.max_graphs:
                    self._cache.popitem(last=False)       # drops the graph and its private pool
            else:
                self._cache.move_to_end(key)
            static_in, graph, flat, rebuild = entry
            for s, t in zip(static_in, inputs):
                s.copy_(t)
            graph.replay()
            self.replays += 1
            return rebuild([t.clone() for t in flat])     # the static outputs are overwritten by the next replay
