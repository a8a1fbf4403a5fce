"""ctypes binding of libptops.so (include/ptops.h) for torch tensors.

This is the only place the package touches the native library.  Every wrapper checks
device / dtype / contiguity (the reference's extensions check nothing,
extensions/chamfer_dist/chamfer.cu:147-171), launches on torch's CURRENT stream (the reference
launches on the default stream) and raises RuntimeError with the library's message
on a non-zero status.

There is no CPU fallback: a missing library or a non-GPU tensor raises.  Tests (and only
tests / the bench's cpu_baseline leg) may install a replacement backend object with
`set_backend_for_testing`, e.g. one that routes to the CPU oracle to exercise the Python
plumbing on a machine without a GPU; the package itself never imports the oracle.
"""
import contextlib
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptops.so")

_lib = None
_test_backend = None


class PtopsError(RuntimeError):
    pass


def _declare(lib):
    c_int, c_void, c_size = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    P = c_void  # raw device pointers travel as integers
    sig = {
        "ptops_abi_version": (c_int, []),
        "ptops_last_error": (ctypes.c_char_p, []),
        "ptops_chamfer_forward_workspace_bytes": (c_size, [c_int] * 3),
        "ptops_chamfer_forward": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, P, P]),
        "ptops_chamfer_forward_ex": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, P, c_int, c_int, P, P]),
        "ptops_chamfer_sweep_available": (c_int, [c_int] * 3),
        "ptops_chamfer_backward": (c_int, [P] * 6 + [c_int] * 3 + [P, P, P]),
        "ptops_chamfer_backward_det_workspace_bytes": (c_size, [c_int] * 3),
        "ptops_chamfer_backward_det": (c_int, [P] * 6 + [c_int] * 3 + [P, P, P, c_size, P]),
        "ptops_chamfer_reduce_workspace_bytes": (c_size, [ctypes.c_longlong, ctypes.c_longlong]),
        "ptops_chamfer_reduce": (c_int, [P, ctypes.c_longlong, P, ctypes.c_longlong, c_int, P, P, P]),
        "ptops_chamfer_reduce_backward": (c_int, [P, ctypes.c_longlong, P, ctypes.c_longlong, c_int, c_int, P, P, P, P, P]),
        "ptops_furthest_point_sample": (c_int, [P, c_int, c_int, c_int, P, P]),
        "ptops_gather_forward": (c_int, [P, P] + [c_int] * 4 + [P, P]),
        "ptops_gather_backward": (c_int, [P, P] + [c_int] * 4 + [P, P]),
        "ptops_knn": (c_int, [P, P] + [c_int] * 4 + [P, P]),
        "ptops_topk_rows": (c_int, [P, ctypes.c_longlong, c_int, c_int, P, P]),
        "ptops_index_points_forward": (c_int, [P, P] + [c_int] * 5 + [P, P]),
        "ptops_index_points_backward": (c_int, [P, P] + [c_int] * 5 + [P, P]),
        "ptops_edge_group_forward": (c_int, [P, P, P] + [c_int] * 5 + [P, P]),
        "ptops_edge_group_backward": (c_int, [P, P] + [c_int] * 5 + [P, P, P]),
        "ptops_gather_max_forward": (c_int, [P, P] + [c_int] * 5 + [P, P, P]),
        "ptops_gather_max_backward": (c_int, [P, P] + [c_int] * 4 + [P, P]),
        "ptops_gather_max_act_forward": (c_int, [P, P, P, ctypes.c_float] + [c_int] * 5 + [P, P, P]),
        "ptops_gather_max_act_backward": (c_int, [P, P, P, ctypes.c_float] + [c_int] * 4 + [P, P, P]),
        "ptops_emd_forward": (c_int, [P, P, c_int, c_int, c_int] + [P] * 12 + [ctypes.c_float, c_int, P]),
        "ptops_emd_backward": (c_int, [P] * 5 + [c_int] * 3 + [P]),
        "ptops_edge_select_blocks": (c_int, [c_int]),
        "ptops_edge_select": (c_int, [P, P, P, P] + [c_int] * 5 + [P, P, P, P, P]),
        "ptops_edge_scatter": (c_int, [P, P, P, P] + [c_int] * 5 + [P, P]),
        "ptops_edge_norm_forward": (c_int, [P] * 4 + [c_int] * 6 + [ctypes.c_float, ctypes.c_float, P, P, P]),
        "ptops_edge_norm_backward_workspace_bytes": (c_size, [c_int] * 6),
        "ptops_edge_norm_backward": (c_int, [P] * 10 + [c_int] * 6 + [ctypes.c_float] + [P] * 5),
        "ptops_colsum_workspace_bytes": (c_size, [c_int, c_int]),
        "ptops_colsum": (c_int, [P, c_int, c_int, P, P, P]),
        "ptops_adamw_row_bytes": (c_int, []),
        "ptops_adamw_chunk_elements": (c_int, []),
        "ptops_adamw_clip_step": (c_int, [P, c_int, P, c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                          P, P, P]),
        "ptops_fold_defer": (c_int, [P, c_int]),
        "ptops_set_option": (c_int, [ctypes.c_char_p, c_int]),
        "ptops_get_option": (c_int, [ctypes.c_char_p]),
        "ptops_fold_flush": (c_int, [P]),
        "ptops_max_tokens_forward": (c_int, [P, c_int, c_int, c_int, P, P, P]),
        "ptops_gemm_f32_split3": (c_int, [P, ctypes.c_longlong, c_int, P, ctypes.c_longlong, c_int, P, ctypes.c_longlong,
                                          c_int, c_int, c_int, P, P]),
        "ptops_gemm_f32_split3_gelu": (c_int, [P, ctypes.c_longlong, P, ctypes.c_longlong, P, ctypes.c_longlong, P,
                                               ctypes.c_longlong, c_int, c_int, c_int, P, P]),
        "ptops_gemm_f32_split3_wgrad_workspace": (ctypes.c_longlong, [c_int, c_int, c_int]),
        "ptops_gemm_f32_split3_wgrad": (c_int, [P, ctypes.c_longlong, P, ctypes.c_longlong, c_int, c_int, c_int, P,
                                                ctypes.c_longlong, P, P, ctypes.c_longlong, P]),
        "ptops_wgrad_problem_bytes": (c_int, []),
        "ptops_wgrad_group_plan": (ctypes.c_longlong, [P, P, c_int, c_int, P, ctypes.c_longlong]),
        "ptops_wgrad_group_workspace_bytes": (ctypes.c_longlong, [P]),
        "ptops_gemm_f32_split3_wgrad_grouped": (c_int, [P, P, c_int, P, P, P, ctypes.c_longlong, P]),
        "ptops_max_tokens_backward": (c_int, [P, P, c_int, c_int, c_int, P, P]),
        "ptops_attention_forward": (c_int, [P, P, P] + [c_int] * 4 + [P, P, P, c_int, c_int, ctypes.c_float, P, P, P]),
        "ptops_layer_norm_max_dim": (c_int, []),
        "ptops_layer_norm_forward": (c_int, [P, P, P, ctypes.c_longlong, c_int, ctypes.c_float, P, P, P, P]),
        "ptops_layer_norm_backward_workspace_bytes": (c_size, [ctypes.c_longlong, c_int]),
        "ptops_add_layer_norm_forward": (c_int, [P, P, P, P, ctypes.c_longlong, c_int, ctypes.c_float, P, P, P, P, P]),
        "ptops_layer_norm_backward": (c_int, [P] * 6 + [ctypes.c_longlong, c_int, P, P, P, P]),
        "ptops_gelu_backward_colsum_workspace_bytes": (c_size, [c_int, c_int]),
        "ptops_gelu_backward_colsum": (c_int, [P, P, c_int, c_int, P, P, P, P]),
        "ptops_batch_norm_rows_workspace_bytes": (c_size, [c_int, c_int]),
        "ptops_batch_norm_rows_forward": (c_int, [P, P, P, c_int, c_int, ctypes.c_float, ctypes.c_float, P, P, P, P, P]),
        "ptops_batch_norm_rows_backward": (c_int, [P] * 5 + [c_int, c_int, ctypes.c_float, P, P, P, P]),
        "ptops_attention_backward_fused_workspace_bytes": (c_size, [c_int] * 4),
        "ptops_attention_backward_fused": (c_int, [P] * 6 + [c_int] * 4 + [P] * 6 + [c_int, c_int, ctypes.c_float] + [P] * 5),
        "ptops_attention_backward": (c_int, [P] * 6 + [c_int] * 4 + [P] * 6 + [c_int, c_int, ctypes.c_float] + [P] * 5),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTS = (
    "ptops_abi_version", "ptops_last_error", "ptops_chamfer_forward_workspace_bytes",
    "ptops_chamfer_forward", "ptops_chamfer_forward_ex", "ptops_chamfer_sweep_available",
    "ptops_chamfer_backward", "ptops_chamfer_backward_det_workspace_bytes", "ptops_chamfer_backward_det",
    "ptops_chamfer_reduce_workspace_bytes", "ptops_chamfer_reduce", "ptops_chamfer_reduce_backward",
    "ptops_furthest_point_sample",
    "ptops_gather_forward", "ptops_gather_backward", "ptops_knn", "ptops_topk_rows", "ptops_index_points_forward",
    "ptops_index_points_backward", "ptops_edge_group_forward", "ptops_edge_group_backward",
    "ptops_gather_max_forward", "ptops_gather_max_backward", "ptops_gather_max_act_forward", "ptops_gather_max_act_backward", "ptops_emd_forward", "ptops_emd_backward",
    "ptops_attention_forward", "ptops_attention_backward", "ptops_attention_backward_fused_workspace_bytes",
    "ptops_attention_backward_fused", "ptops_colsum_workspace_bytes", "ptops_colsum",
    "ptops_fold_defer", "ptops_fold_flush", "ptops_set_option", "ptops_get_option", "ptops_adamw_row_bytes", "ptops_adamw_chunk_elements", "ptops_adamw_clip_step",
    "ptops_max_tokens_forward", "ptops_max_tokens_backward", "ptops_gemm_f32_split3",
    "ptops_gemm_f32_split3_wgrad_workspace", "ptops_gemm_f32_split3_wgrad", "ptops_gemm_f32_split3_gelu",
    "ptops_wgrad_problem_bytes", "ptops_wgrad_group_plan", "ptops_wgrad_group_workspace_bytes",
    "ptops_gemm_f32_split3_wgrad_grouped",
    "ptops_edge_select_blocks", "ptops_edge_select", "ptops_edge_scatter",
    "ptops_edge_norm_forward", "ptops_edge_norm_backward_workspace_bytes", "ptops_edge_norm_backward",
    "ptops_batch_norm_rows_workspace_bytes", "ptops_batch_norm_rows_forward", "ptops_batch_norm_rows_backward",
    "ptops_gelu_backward_colsum_workspace_bytes", "ptops_gelu_backward_colsum",
    "ptops_layer_norm_max_dim", "ptops_layer_norm_forward", "ptops_add_layer_norm_forward",
    "ptops_layer_norm_backward_workspace_bytes",
    "ptops_layer_norm_backward",
)


def load():
    """dlopen libptops.so (built in-tree by imgadapointr_amd/csrc/Makefile); raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtopsError(
                f"{LIB_PATH} is missing: build it with `make -C imgadapointr_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        _declare(lib)
        _lib = lib
    return _lib


def set_backend_for_testing(backend):
    """Install (or, with None, remove) a replacement for the functions of this module.

    Test infrastructure only: `backend` must provide the same function names as this module
    (chamfer_forward, ..., gather_max_backward) taking/returning torch tensors."""
    global _test_backend
    _test_backend = backend


def backend_for_testing():
    return _test_backend


class _on:
    """`with torch.cuda.device(dev)` without the per-use argument parsing (two C calls; the wrappers below run ~550
    times per train step)."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        self.idx = dev.index if dev.index is not None else -1

    def __enter__(self):
        self.prev = torch._C._cuda_exchangeDevice(self.idx) if self.idx >= 0 else -1
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch._C._cuda_maybeExchangeDevice(self.prev)
        return False


class _RawStream:
    """The current HIP stream of a device as its raw handle (`.cuda_stream`, like torch.cuda.Stream): one C call instead
    of building a torch.cuda.Stream object per kernel launch (2.7 us each, 550 times per train step)."""
    __slots__ = ("cuda_stream",)

    def __init__(self, handle):
        self.cuda_stream = handle


def _current_stream(dev=None):
    idx = dev.index if (dev is not None and dev.index is not None) else torch.cuda.current_device()
    return _RawStream(torch._C._cuda_getCurrentRawStream(idx))


def _stream(t):
    return ctypes.c_void_p(_current_stream(t.device).cuda_stream)


def _check(t, name, dtype, ndim):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise PtopsError(f"{name}: tensor is on {t.device}; libptops runs on the GPU only "
                         "(no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _ptr(t):
    return t if isinstance(t, ctypes.c_void_p) else ctypes.c_void_p(t.data_ptr())


_entry_points = {}


def _call(fn_name, *args):
    fn = _entry_points.get(fn_name)
    if fn is None:
        fn = _entry_points[fn_name] = getattr(load(), fn_name)
    rc = fn(*args)
    if rc != 0:
        raise PtopsError(f"{fn_name} failed ({rc}): {load().ptops_last_error().decode()}")


def _same_device(*ts):
    d = ts[0].device
    for t in ts[1:]:
        if t.device != d:
            raise ValueError(f"tensors on different devices: {d} vs {t.device}")
    return d


# ---- HIP-graph capture of a whole step (utils/graph_runner.py::GraphedTrainStep) ------------------------------------
# A captured launch sequence is replayed without running this module again, so everything the host decides per call must
# be frozen for the life of the graph, and every buffer whose address is baked into it must be owned by the graph:
#   * `private_state()` gives the warm-up steps and the capture their OWN workspace tables (fold arena, column-sum /
#     weight-gradient / Chamfer-reduction workspaces, device problem tables); the caller keeps the returned object alive
#     as long as the graph, and nothing the eager path does afterwards can free or recycle those buffers;
#   * inside `capturing()` the Chamfer policy neither probes nor reads statistics back (its current per-shape choice is
#     used as it stands), and host tables that a captured copy node reads (the pointer table of the grouped weight
#     gradients) are pinned and kept alive by the same object.
_capture = None               # None, or the PrivateState of the capture in progress


class PrivateState:
    """The workspace tables of one graphed step (see above).  A context manager: inside, this module's tables are these."""
    _NAMES = ("_colsum_ws", "_fold_arena", "_wgrad_ws", "_wgrad_tables", "_chamfer_reduce_ws")

    def __init__(self):
        self.tables = {n: {} for n in self._NAMES}
        self.keepalive = []            # pinned host tables, plans: whatever a captured node reads
        self._saved = None
        # Pinned staging for the host tables a captured copy node reads (pointer tables, plans): allocated by
        # `new_pinned_pool()` BEFORE each capture - hipHostMalloc is not permitted while a stream of the process is
        # capturing, and a flush whose queue differs from every warm-up flush would otherwise need a fresh pinned block in
        # the middle of the capture.  One pool per capture: the graph's cache entry keeps it, an evicted graph frees it.
        self._pinned = None
        self._pinned_used = 0
        self._wgrad_side = {}          # device index -> side stream of the weight-gradient branch (wgrad_flush)
        self._wgrad_forked = None      # (backward stream, side stream) while flushes are outstanding on the branch

    def fork_wgrad_branch(self, dev, queue):
        """The side stream of the weight-gradient branch, made to wait for the work queued so far on the current (backward)
        stream; the operands of `queue` are marked as used there."""
        side = self._wgrad_side.get(dev.index)
        if side is None:
            side = self._wgrad_side[dev.index] = torch.cuda.Stream(device=dev)
        main = _deferral_stream(dev)              # the stream the operands were produced on (not necessarily the caller's)
        side.wait_stream(main)
        for q in queue:
            q[0].record_stream(side)
            q[1].record_stream(side)
        self._wgrad_forked = (main, side)
        return side

    def join_wgrad_branch(self):
        if self._wgrad_forked is not None:
            main, side = self._wgrad_forked
            main.wait_stream(side)
            self._wgrad_forked = None

    def new_pinned_pool(self, nbytes=1 << 20):
        self._pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self._pinned_used = 0
        return self._pinned

    def pinned_copy(self, array):
        """A pinned-memory copy of a numpy array (kept for the life of this object) -> uint8 tensor view of its bytes."""
        import numpy as np
        raw = np.ascontiguousarray(array).view(np.uint8).reshape(-1)
        n = raw.size
        start = (self._pinned_used + 63) & ~63
        if self._pinned is None or start + n > self._pinned.numel():
            raise PtopsError("private_state: pinned staging pool exhausted")
        self._pinned_used = start + n
        out = self._pinned[start:start + n]
        out.numpy()[:] = raw
        return out

    def __enter__(self):
        g = globals()
        assert self._saved is None, "PrivateState is not re-entrant"
        if _fold_deferred or _fold_keepalive or _wgrad_queue:
            raise PtopsError("private_state: entered while gradients are queued (inside deferred_folds)")
        self._saved = {n: g[n] for n in self._NAMES}
        for n in self._NAMES:
            g[n] = self.tables[n]
        return self

    def __exit__(self, *exc):
        g = globals()
        for n in self._NAMES:
            g[n] = self._saved[n]
        self._saved = None
        return False


def private_state():
    return PrivateState()


class capturing:
    """Marks a stream capture in progress (entered inside `torch.cuda.graph`, inside the PrivateState `state`)."""

    def __init__(self, state):
        self.state = state

    def __enter__(self):
        global _capture
        assert _capture is None
        _capture = self.state
        return self

    def __exit__(self, *exc):
        global _capture
        _capture = None
        return False


# ------------------------------------------------------------------------------------------
CHAMFER_AUTO, CHAMFER_BRUTE, CHAMFER_SWEEP, CHAMFER_MFMA = 0, 1, 2, 3
CHAMFER_MFMA_MIN_PAIRS = 1 << 32        # PTOPS_CHAMFER_MFMA_MIN_PAIRS of include/ptops.h


class _ChamferPolicy:
    """Per-shape, per-direction choice between the three exact Chamfer algorithms (bit-identical results).

    * pruned sweep (both clouds 4096..16384 points): ~3x faster than brute force when nearest neighbours are close
      (trained predictions, well-spread clouds), useless when they are not (uniform queries against a prediction
      collapsed into a few clusters);
    * MFMA filter (launches of >= 2^32 pair evaluations, e.g. B = 32 at 16384 x 16384): sees every pair on the matrix
      pipe, 2x faster than brute force whatever the geometry (0.95 vs 1.9 ms), unless masses of exactly tied distances
      (lattices, duplicated points) force queries into their exact scan; below that size its prep pass and exact
      phase cost more than they save;
    * brute force: the VALU kernel.

    Every swept / filtered direction accumulates {work done exactly, work a full scan would do} on the device; the
    counters are copied to pinned host memory without synchronising and looked at on a LATER call.  A sweep that still
    evaluated more than `sweep_threshold` of the chunks is demoted to the next rung, an MFMA filter that re-evaluated
    more than `mfma_threshold` of the superblocks exactly is demoted to brute force, each for the next `hold` calls of
    the shape; then the preferred algorithm is probed again."""
    sweep_threshold = 0.35
    mfma_threshold = 0.03
    hold = 50
    mfma_min_pairs = CHAMFER_MFMA_MIN_PAIRS

    def __init__(self):
        self.state = {}

    def _ladder(self, sweep_ok, pairs):
        ladder = [CHAMFER_SWEEP] if sweep_ok else []
        if pairs >= self.mfma_min_pairs:
            ladder.append(CHAMFER_MFMA)
        return ladder + [CHAMFER_BRUTE]

    def choose(self, key, sweep_ok, pairs):
        """-> ([algorithm of direction 0, of direction 1], state).  pairs = B * n * m, the pair evaluations of a direction."""
        st = self.state.setdefault(key, dict(level=[0, 0], left=[0, 0], pending=None))
        ladders = [self._ladder(sweep_ok, pairs) for d in range(2)]
        pend = st["pending"]
        if pend is not None and pend[1].query():
            vals, used = pend[0].tolist(), pend[2]
            st["pending"] = None
            for d in range(2):
                ev, tot = vals[2 * d], vals[2 * d + 1]
                if tot <= 0 or used[d] not in ladders[d]:
                    continue
                limit = self.sweep_threshold if used[d] == CHAMFER_SWEEP else self.mfma_threshold
                if used[d] != CHAMFER_BRUTE and ev / tot > limit:
                    st["level"][d] = ladders[d].index(used[d]) + 1
                    st["left"][d] = self.hold
        algos = []
        for d in range(2):
            if st["left"][d] > 0:
                st["left"][d] -= 1
            else:
                st["level"][d] = 0
            algos.append(ladders[d][min(st["level"][d], len(ladders[d]) - 1)])
        return algos, st

    def frozen(self, key, sweep_ok, pairs):
        """The choice `choose` would make now, without touching the state: what a captured launch sequence bakes in."""
        st = self.state.get(key, dict(level=[0, 0], left=[0, 0]))
        ladder = self._ladder(sweep_ok, pairs)
        return [ladder[min(st["level"][d] if st["left"][d] > 0 else 0, len(ladder) - 1)] for d in range(2)]

    @staticmethod
    def record(st, stats_dev, algos):
        if st is None or st["pending"] is not None:
            return
        host = torch.empty(4, dtype=torch.int64, pin_memory=True)
        host.copy_(stats_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["pending"] = (host, ev, tuple(algos))


_chamfer_policy = _ChamferPolicy()


chamfer_algorithm_log = {}   # (B, n, m) -> algorithms of the most recent launch, for logs and bench.py


def chamfer_forward(xyz1, xyz2, algorithm=CHAMFER_AUTO):
    """chamfer.forward (extensions/chamfer_dist/chamfer_cuda.cpp:22-25).  `algorithm`: CHAMFER_AUTO
    (adaptive per direction, see _ChamferPolicy), CHAMFER_BRUTE, CHAMFER_SWEEP, CHAMFER_MFMA, or a pair of those for
    (xyz1->xyz2, xyz2->xyz1) — the results are identical."""
    if _test_backend is not None:
        return _test_backend.chamfer_forward(xyz1, xyz2)
    _check(xyz1, "xyz1", torch.float32, 3)
    _check(xyz2, "xyz2", torch.float32, 3)
    dev = _same_device(xyz1, xyz2)
    B, n, c = xyz1.shape
    B2, m, c2 = xyz2.shape
    if c != 3 or c2 != 3 or B != B2:
        raise ValueError(f"chamfer_forward: expected [B,n,3] and [B,m,3], got {tuple(xyz1.shape)} {tuple(xyz2.shape)}")
    lib = load()
    algos = list(algorithm) if isinstance(algorithm, (tuple, list)) else [algorithm, algorithm]
    with _on(dev):
        dist1 = torch.empty((B, n), dtype=torch.float32, device=dev)
        dist2 = torch.empty((B, m), dtype=torch.float32, device=dev)
        idx1 = torch.empty((B, n), dtype=torch.int32, device=dev)
        idx2 = torch.empty((B, m), dtype=torch.int32, device=dev)
        if B == 0 or n == 0 or m == 0:
            algos = [CHAMFER_BRUTE, CHAMFER_BRUTE]
        wsb = lib.ptops_chamfer_forward_workspace_bytes(B, n, m)
        ws = torch.empty(((wsb + 15) // 16 * 2,), dtype=torch.int64, device=dev) if wsb else None
        stats = st = None
        if algos == [CHAMFER_AUTO, CHAMFER_AUTO] and _capture is not None:
            algos = _chamfer_policy.frozen((dev.index, B, n, m), bool(lib.ptops_chamfer_sweep_available(B, n, m)), B * n * m)
        elif algos == [CHAMFER_AUTO, CHAMFER_AUTO]:
            algos, st = _chamfer_policy.choose((dev.index, B, n, m), bool(lib.ptops_chamfer_sweep_available(B, n, m)),
                                               B * n * m)
            if st["pending"] is None and any(a in (CHAMFER_SWEEP, CHAMFER_MFMA) for a in algos):
                stats = torch.zeros(4, dtype=torch.int64, device=dev)
        _call("ptops_chamfer_forward_ex", _ptr(xyz1), _ptr(xyz2), B, n, m, _ptr(dist1), _ptr(dist2),
              _ptr(idx1), _ptr(idx2), _ptr(ws) if ws is not None else None, int(algos[0]), int(algos[1]),
              _ptr(stats) if stats is not None else None, _stream(xyz1))
        if stats is not None:
            _chamfer_policy.record(st, stats, algos)
        chamfer_algorithm_log[(B, n, m)] = tuple(int(a) for a in algos)
    return dist1, dist2, idx1, idx2


_chamfer_reduce_ws = {}      # (device index, stream) -> zero-armed workspace of ptops_chamfer_reduce


def chamfer_reduce(dist1, dist2, take_sqrt):
    """-> [4] = mean(f(dist1)), mean(f(dist2)), their sum, half their sum (f = sqrt or identity) in one launch
    (csrc/chamfer.hip: the reductions of extensions/chamfer_dist/__init__.py:28-103; deterministic)."""
    dev = _same_device(dist1, dist2)
    for name, t in (("dist1", dist1), ("dist2", dist2)):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError(f"chamfer_reduce: {name} must be a contiguous float32 tensor")
    n1, n2 = dist1.numel(), dist2.numel()
    with _on(dev):
        stream = _current_stream(dev)
        need = int(load().ptops_chamfer_reduce_workspace_bytes(n1, n2))
        key = (dev.index, stream.cuda_stream)
        ws = _chamfer_reduce_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.zeros((max(need, 4096),), dtype=torch.uint8, device=dev)     # the counter starts (and stays) at zero
            _chamfer_reduce_ws[key] = ws
        out = torch.empty((4,), dtype=torch.float32, device=dev)
        _call("ptops_chamfer_reduce", _ptr(dist1), n1, _ptr(dist2), n2, int(bool(take_sqrt)), _ptr(out), _ptr(ws),
              ctypes.c_void_p(stream.cuda_stream))
    return out


def chamfer_reduce_backward(dist1, dist2, take_sqrt, which, go1, go2=None):
    """Gradients of dist1 / dist2 for the gradient of one output of `chamfer_reduce`: `which` 2 / 3 -> of out[2] / out[3]
    (go1, a one-element tensor), 0 -> of out[0] and out[1] (go1, go2; None = no gradient)."""
    dev = _same_device(dist1, dist2)
    with _on(dev):
        g1, g2 = torch.empty_like(dist1), torch.empty_like(dist2)
        _call("ptops_chamfer_reduce_backward", _ptr(dist1), dist1.numel(), _ptr(dist2), dist2.numel(), int(bool(take_sqrt)),
              int(which), _ptr(go1) if go1 is not None else None, _ptr(go2) if go2 is not None else None, _ptr(g1), _ptr(g2),
              _stream(dist1))
    return g1, g2


# True: scatter-add backward passes with a fixed summation order (bit-identical gradients from run to run) where the
# library has one: Chamfer (max(n, m) <= 16384).  False: the reference's atomicAdd form everywhere.
DETERMINISTIC_BACKWARD = os.environ.get("PTOPS_DETERMINISTIC_BACKWARD", "1") != "0"


def chamfer_backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2, deterministic=None):
    """chamfer.backward (extensions/chamfer_dist/chamfer_cuda.cpp:27-34).  `deterministic` (default: the module
    switch): own term + neighbour terms by ascending index instead of atomicAdd arrival order."""
    if _test_backend is not None:
        return _test_backend.chamfer_backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2)
    _check(xyz1, "xyz1", torch.float32, 3)
    _check(xyz2, "xyz2", torch.float32, 3)
    _check(idx1, "idx1", torch.int32, 2)
    _check(idx2, "idx2", torch.int32, 2)
    _check(grad_dist1, "grad_dist1", torch.float32, 2)
    _check(grad_dist2, "grad_dist2", torch.float32, 2)
    dev = _same_device(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    if tuple(idx1.shape) != (B, n) or tuple(idx2.shape) != (B, m) or \
            tuple(grad_dist1.shape) != (B, n) or tuple(grad_dist2.shape) != (B, m):
        raise ValueError("chamfer_backward: idx/grad shapes do not match the clouds")
    with _on(dev):
        g1 = torch.empty_like(xyz1)
        g2 = torch.empty_like(xyz2)
        det = DETERMINISTIC_BACKWARD if deterministic is None else deterministic
        if det and B > 0 and 0 < n <= 16384 and 0 < m <= 16384:
            stream = _current_stream(dev)
            need = load().ptops_chamfer_backward_det_workspace_bytes(B, n, m)
            ws = _scratch(dev, stream, need)
            _call("ptops_chamfer_backward_det", _ptr(xyz1), _ptr(xyz2), _ptr(idx1), _ptr(idx2), _ptr(grad_dist1),
                  _ptr(grad_dist2), B, n, m, _ptr(g1), _ptr(g2), _ptr(ws), ctypes.c_size_t(need),
                  ctypes.c_void_p(stream.cuda_stream))
        else:
            _call("ptops_chamfer_backward", _ptr(xyz1), _ptr(xyz2), _ptr(idx1), _ptr(idx2),
                  _ptr(grad_dist1), _ptr(grad_dist2), B, n, m, _ptr(g1), _ptr(g2), _stream(xyz1))
    return g1, g2


def furthest_point_sample(xyz, npoint):
    """pointnet2_utils.furthest_point_sample: xyz [B,N,3] -> int32 [B,npoint]."""
    if _test_backend is not None:
        return _test_backend.furthest_point_sample(xyz, npoint)
    _check(xyz, "xyz", torch.float32, 3)
    B, N, c = xyz.shape
    if c != 3:
        raise ValueError(f"furthest_point_sample: expected [B,N,3], got {tuple(xyz.shape)}")
    with _on(xyz.device):
        idx = torch.empty((B, int(npoint)), dtype=torch.int32, device=xyz.device)
        _call("ptops_furthest_point_sample", _ptr(xyz), B, N, int(npoint), _ptr(idx), _stream(xyz))
    return idx


def gather_forward(feat, idx):
    """gather_operation forward: feat [B,C,N], idx int32 [B,npoint] -> [B,C,npoint]."""
    if _test_backend is not None:
        return _test_backend.gather_forward(feat, idx)
    _check(feat, "features", torch.float32, 3)
    _check(idx, "idx", torch.int32, 2)
    dev = _same_device(feat, idx)
    B, C, N = feat.shape
    npoint = idx.shape[1]
    with _on(dev):
        out = torch.empty((B, C, npoint), dtype=torch.float32, device=dev)
        _call("ptops_gather_forward", _ptr(feat), _ptr(idx), B, C, N, npoint, _ptr(out), _stream(feat))
    return out


def gather_backward(grad_out, idx, N):
    if _test_backend is not None:
        return _test_backend.gather_backward(grad_out, idx, N)
    _check(grad_out, "grad_out", torch.float32, 3)
    _check(idx, "idx", torch.int32, 2)
    dev = _same_device(grad_out, idx)
    B, C, npoint = grad_out.shape
    with _on(dev):
        g = torch.empty((B, C, int(N)), dtype=torch.float32, device=dev)
        _call("ptops_gather_backward", _ptr(grad_out), _ptr(idx), B, C, int(N), npoint, _ptr(g),
              _stream(grad_out))
    return g


def knn(k, xyz, new_xyz):
    """Fused knn_point: xyz [B,N,3] candidates, new_xyz [B,S,3] queries -> int32 [B,S,k]."""
    if _test_backend is not None:
        return _test_backend.knn(k, xyz, new_xyz)
    _check(xyz, "xyz", torch.float32, 3)
    _check(new_xyz, "new_xyz", torch.float32, 3)
    dev = _same_device(xyz, new_xyz)
    B, N, c = xyz.shape
    B2, S, c2 = new_xyz.shape
    if c != 3 or c2 != 3 or B != B2:
        raise ValueError(f"knn: expected [B,N,3] and [B,S,3], got {tuple(xyz.shape)} {tuple(new_xyz.shape)}")
    with _on(dev):
        idx = torch.empty((B, S, int(k)), dtype=torch.int32, device=dev)
        _call("ptops_knn", _ptr(xyz), _ptr(new_xyz), B, N, S, int(k), _ptr(idx), _stream(xyz))
    return idx


def topk_rows(x, k):
    """x [...,N] float32 -> int64 [...,k]: the k largest per row by (value descending, column ascending);
    `x.topk(k, dim=-1)[1]` with a defined order of equal values (GDANet's feature-space kNN)."""
    if _test_backend is not None:
        return _test_backend.topk_rows(x, k)
    _check(x, "x", torch.float32, x.dim())
    N = x.size(-1)
    k = int(k)
    if not (x.dim() >= 1 and 1 <= k <= N <= 2048):
        raise ValueError(f"topk_rows: need 1 <= k <= N <= 2048, got k={k} for {tuple(x.shape)}")
    with _on(x.device):
        idx = torch.empty(tuple(x.shape[:-1]) + (k,), dtype=torch.int64, device=x.device)
        _call("ptops_topk_rows", _ptr(x), x.numel() // N, N, k, _ptr(idx), _stream(x))
    return idx


def index_points_forward(points, idx):
    """points [B,N,C], idx int32 [B,S,k] -> [B,S,k,C]."""
    if _test_backend is not None:
        return _test_backend.index_points_forward(points, idx)
    _check(points, "points", torch.float32, 3)
    _check(idx, "idx", torch.int32, 3)
    dev = _same_device(points, idx)
    B, N, C = points.shape
    _, S, k = idx.shape
    with _on(dev):
        out = torch.empty((B, S, k, C), dtype=torch.float32, device=dev)
        _call("ptops_index_points_forward", _ptr(points), _ptr(idx), B, N, C, S, k, _ptr(out),
              _stream(points))
    return out


def index_points_backward(grad_out, idx, N):
    if _test_backend is not None:
        return _test_backend.index_points_backward(grad_out, idx, N)
    _check(grad_out, "grad_out", torch.float32, 4)
    _check(idx, "idx", torch.int32, 3)
    dev = _same_device(grad_out, idx)
    B, S, k, C = grad_out.shape
    with _on(dev):
        g = torch.empty((B, int(N), C), dtype=torch.float32, device=dev)
        _call("ptops_index_points_backward", _ptr(grad_out), _ptr(idx), B, int(N), C, S, k, _ptr(g),
              _stream(grad_out))
    return g


def edge_group_forward(x_k, x_q, idx):
    """x_k [B,C,Nk], x_q [B,C,Nq], idx int32 [B,Nq,k] -> [B,2C,Nq,k]."""
    if _test_backend is not None:
        return _test_backend.edge_group_forward(x_k, x_q, idx)
    _check(x_k, "x_k", torch.float32, 3)
    _check(x_q, "x_q", torch.float32, 3)
    _check(idx, "idx", torch.int32, 3)
    dev = _same_device(x_k, x_q, idx)
    B, C, Nk = x_k.shape
    Nq = x_q.shape[2]
    k = idx.shape[2]
    if tuple(x_q.shape[:2]) != (B, C) or tuple(idx.shape[:2]) != (B, Nq):
        raise ValueError("edge_group_forward: inconsistent shapes")
    with _on(dev):
        out = torch.empty((B, 2 * C, Nq, k), dtype=torch.float32, device=dev)
        _call("ptops_edge_group_forward", _ptr(x_k), _ptr(x_q), _ptr(idx), B, C, Nk, Nq, k, _ptr(out),
              _stream(x_k))
    return out


def edge_group_backward(grad_out, idx, Nk):
    if _test_backend is not None:
        return _test_backend.edge_group_backward(grad_out, idx, Nk)
    _check(grad_out, "grad_out", torch.float32, 4)
    _check(idx, "idx", torch.int32, 3)
    dev = _same_device(grad_out, idx)
    B, C2, Nq, k = grad_out.shape
    C = C2 // 2
    with _on(dev):
        gk = torch.empty((B, C, int(Nk)), dtype=torch.float32, device=dev)
        gq = torch.empty((B, C, Nq), dtype=torch.float32, device=dev)
        _call("ptops_edge_group_backward", _ptr(grad_out), _ptr(idx), B, C, int(Nk), Nq, k, _ptr(gk),
              _ptr(gq), _stream(grad_out))
    return gk, gq


def gather_max_forward(p, idx):
    """p [B,N,C], idx int32 [B,S,k] -> (max [B,S,C], arg int32 [B,S,C])."""
    if _test_backend is not None:
        return _test_backend.gather_max_forward(p, idx)
    _check(p, "p", torch.float32, 3)
    _check(idx, "idx", torch.int32, 3)
    dev = _same_device(p, idx)
    B, N, C = p.shape
    _, S, k = idx.shape
    with _on(dev):
        out = torch.empty((B, S, C), dtype=torch.float32, device=dev)
        arg = torch.empty((B, S, C), dtype=torch.int32, device=dev)
        _call("ptops_gather_max_forward", _ptr(p), _ptr(idx), B, N, C, S, k, _ptr(out), _ptr(arg),
              _stream(p))
    return out, arg


def gather_max_backward(grad_out, arg, N):
    if _test_backend is not None:
        return _test_backend.gather_max_backward(grad_out, arg, N)
    _check(grad_out, "grad_out", torch.float32, 3)
    _check(arg, "arg", torch.int32, 3)
    dev = _same_device(grad_out, arg)
    B, S, C = grad_out.shape
    with _on(dev):
        g = torch.empty((B, int(N), C), dtype=torch.float32, device=dev)
        _call("ptops_gather_max_backward", _ptr(grad_out), _ptr(arg), B, int(N), C, S, _ptr(g),
              _stream(grad_out))
    return g


def gather_max_act_forward(p, idx, add, slope):
    """p [B,N,C], idx int32 [B,S,k], add [B,S,C] -> (LeakyReLU(max_j p[idx] + add, slope) [B,S,C], arg int32 [B,S,C])."""
    _check(p, "p", torch.float32, 3)
    _check(idx, "idx", torch.int32, 3)
    _check(add, "add", torch.float32, 3)
    dev = _same_device(p, idx, add)
    B, N, C = p.shape
    S, k = idx.shape[1], idx.shape[2]
    if idx.shape[0] != B or tuple(add.shape) != (B, S, C):
        raise ValueError("gather_max_act_forward: inconsistent shapes")
    with _on(dev):
        out = torch.empty((B, S, C), dtype=torch.float32, device=dev)
        arg = torch.empty((B, S, C), dtype=torch.int32, device=dev)
        _call("ptops_gather_max_act_forward", _ptr(p), _ptr(idx), _ptr(add), float(slope), B, N, C, S, k, _ptr(out),
              _ptr(arg), _stream(p))
    return out, arg


def gather_max_act_backward(grad_out, out, arg, slope, N):
    """-> (grad_p [B,N,C], grad_add [B,S,C]) of gather_max_act_forward; N <= 1024."""
    _check(grad_out, "grad_out", torch.float32, 3)
    _check(out, "out", torch.float32, 3)
    _check(arg, "arg", torch.int32, 3)
    dev = _same_device(grad_out, out, arg)
    B, S, C = grad_out.shape
    with _on(dev):
        gp = torch.empty((B, int(N), C), dtype=torch.float32, device=dev)
        ga = torch.empty((B, S, C), dtype=torch.float32, device=dev)
        _call("ptops_gather_max_act_backward", _ptr(grad_out), _ptr(out), _ptr(arg), float(slope), B, int(N), C, S, _ptr(gp),
              _ptr(ga), _stream(grad_out))
    return gp, ga


def emd_forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments,
                unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx, eps, iters):
    """emd.forward (extensions/emd/emd.cpp:6-22): caller-allocated outputs / scratch, mutated in place."""
    if _test_backend is not None:
        return _test_backend.emd_forward(xyz1, xyz2, dist, assignment, price, assignment_inv, bid,
                                         bid_increments, max_increments, unass_idx, unass_cnt, unass_cnt_sum,
                                         cnt_tmp, max_idx, eps, iters)
    _check(xyz1, "xyz1", torch.float32, 3)
    _check(xyz2, "xyz2", torch.float32, 3)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    for t, name, dt, numel in ((dist, "dist", torch.float32, B * n), (assignment, "assignment", torch.int32, B * n),
                               (price, "price", torch.float32, B * m),
                               (assignment_inv, "assignment_inv", torch.int32, B * m),
                               (bid, "bid", torch.int32, B * n), (bid_increments, "bid_increments", torch.float32, B * n),
                               (max_increments, "max_increments", torch.float32, B * m),
                               (unass_idx, "unass_idx", torch.int32, B * n), (unass_cnt, "unass_cnt", torch.int32, B),
                               (unass_cnt_sum, "unass_cnt_sum", torch.int32, B), (cnt_tmp, "cnt_tmp", torch.int32, B),
                               (max_idx, "max_idx", torch.int32, B * m)):
        _check(t, name, dt, t.dim())
        if t.numel() < numel:
            raise ValueError(f"emd_forward: {name} has {t.numel()} elements, needs {numel}")
    _same_device(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments,
                 unass_idx, unass_cnt, unass_cnt_sum, cnt_tmp, max_idx)
    with _on(xyz1.device):
        _call("ptops_emd_forward", _ptr(xyz1), _ptr(xyz2), B, n, m, _ptr(dist), _ptr(assignment), _ptr(price),
              _ptr(assignment_inv), _ptr(bid), _ptr(bid_increments), _ptr(max_increments), _ptr(unass_idx),
              _ptr(unass_cnt), _ptr(unass_cnt_sum), _ptr(cnt_tmp), _ptr(max_idx), ctypes.c_float(float(eps)),
              int(iters), _stream(xyz1))
    return 1


def emd_backward(xyz1, xyz2, gradxyz, graddist, idx):
    """emd.backward (extensions/emd/emd.cpp:24-27): accumulates into the caller's zeroed gradxyz."""
    if _test_backend is not None:
        return _test_backend.emd_backward(xyz1, xyz2, gradxyz, graddist, idx)
    _check(xyz1, "xyz1", torch.float32, 3)
    _check(xyz2, "xyz2", torch.float32, 3)
    _check(gradxyz, "gradxyz", torch.float32, 3)
    _check(graddist, "graddist", torch.float32, 2)
    _check(idx, "idx", torch.int32, 2)
    _same_device(xyz1, xyz2, gradxyz, graddist, idx)
    B, n, _ = xyz1.shape
    with _on(xyz1.device):
        _call("ptops_emd_backward", _ptr(xyz1), _ptr(xyz2), _ptr(gradxyz), _ptr(graddist), _ptr(idx), B, n,
              xyz2.shape[1], _stream(xyz1))
    return 1


def _bhnd_strides(t, name):
    """t [B,H,N,64] float32 CUDA view with a contiguous last dim -> ctypes array of (sb, sh, sn)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        raise PtopsError(f"{name}: expected a float32 GPU tensor [B,H,N,64], got {type(t)} "
                         f"{getattr(t, 'shape', None)} on {getattr(t, 'device', None)}")
    if t.size(3) != 64 or t.stride(3) != 1:
        raise ValueError(f"{name}: head dim must be 64 and contiguous, got shape {tuple(t.shape)} stride {t.stride()}")
    return (ctypes.c_longlong * 3)(t.stride(0), t.stride(1), t.stride(2))


def attention_supported(q, k, split_q, limit_k):
    """True when the fp32 MFMA attention kernel covers this call (else the caller uses SDPA)."""
    return (q.is_cuda and q.dtype == torch.float32 and q.dim() == 4 and q.size(3) == 64 and k.size(3) == 64
            and q.size(2) % 32 == 0 and k.size(2) % 32 == 0 and split_q % 32 == 0 and limit_k % 32 == 0
            and q.stride(3) == 1 and k.stride(3) == 1 and all(s % 4 == 0 for s in q.stride()[:3] + k.stride()[:3])
            and q.data_ptr() % 16 == 0 and k.data_ptr() % 16 == 0)


def attention_forward(q, k, v, split_q, limit_k, scale):
    """q [B,H,Nq,64], k/v [B,H,Nk,64] (strided views allowed) -> (out [B,Nq,H,64], lse [B,H,Nq])."""
    if _test_backend is not None:
        return _test_backend.attention_forward(q, k, v, split_q, limit_k, scale)
    qs, ks, vs = _bhnd_strides(q, "q"), _bhnd_strides(k, "k"), _bhnd_strides(v, "v")
    dev = _same_device(q, k, v)
    B, H, Nq, _ = q.shape
    Nk = k.size(2)
    with _on(dev):
        out = torch.empty((B, Nq, H, 64), dtype=torch.float32, device=dev)
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=dev)
        _call("ptops_attention_forward", _ptr(q), _ptr(k), _ptr(v), B, H, Nq, Nk, qs, ks, vs, int(split_q),
              int(limit_k), ctypes.c_float(float(scale)), _ptr(out), _ptr(lse), _stream(q))
    return out, lse


def attention_backward(q, k, v, out, dout, lse, split_q, limit_k, scale, dq, dk, dv, fused=True):
    """Gradients of attention_forward written into dq [B,H,Nq,64] / dk, dv [B,H,Nk,64] (strided views, e.g.
    the slices of one packed allocation).  out/dout [B,Nq,H,64] contiguous, lse [B,H,Nq] from the forward.
    fused=True: single-pass kernel (5 GEMMs); False: the dq + dk/dv pair (7 GEMMs)."""
    qs, ks, vs = _bhnd_strides(q, "q"), _bhnd_strides(k, "k"), _bhnd_strides(v, "v")
    dqs, dks, dvs = _bhnd_strides(dq, "dq"), _bhnd_strides(dk, "dk"), _bhnd_strides(dv, "dv")
    dev = _same_device(q, k, v, out, dout, lse, dq, dk, dv)
    B, H, Nq, _ = q.shape
    Nk = k.size(2)
    if tuple(out.shape) != (B, Nq, H, 64) or tuple(dout.shape) != (B, Nq, H, 64) or tuple(lse.shape) != (B, H, Nq):
        raise ValueError("attention_backward: out/dout must be [B,Nq,H,64] and lse [B,H,Nq]")
    if tuple(dq.shape) != tuple(q.shape) or tuple(dk.shape) != tuple(k.shape) or tuple(dv.shape) != tuple(v.shape):
        raise ValueError("attention_backward: gradient views must have the shapes of q / k / v")
    if not (out.is_contiguous() and dout.is_contiguous() and lse.is_contiguous()):
        raise ValueError("attention_backward: out, dout and lse must be contiguous")
    with _on(dev):
        if fused:
            stream = _current_stream(dev)
            ws = _scratch(dev, stream, max(load().ptops_attention_backward_fused_workspace_bytes(B, H, Nq, Nk), 16))
            _call("ptops_attention_backward_fused", _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse), B, H, Nq,
                  Nk, qs, ks, vs, dqs, dks, dvs, int(split_q), int(limit_k), ctypes.c_float(float(scale)), _ptr(dq),
                  _ptr(dk), _ptr(dv), _ptr(ws), ctypes.c_void_p(stream.cuda_stream))
            return
        delta = torch.empty((B, H, Nq), dtype=torch.float32, device=dev)
        _call("ptops_attention_backward", _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse), B, H, Nq, Nk,
              qs, ks, vs, dqs, dks, dvs, int(split_q), int(limit_k), ctypes.c_float(float(scale)), _ptr(dq), _ptr(dk),
              _ptr(dv), _ptr(delta), _stream(q))


_colsum_ws = {}


# ---- deferred folds (include/ptops.h): the tiny final launches of the column-sum producers, batched -----------------
_fold_deferred = False
_fold_stream = None           # (device, raw stream handle) deferral is switched on for
_fold_keepalive = []          # workspaces / outputs of queued folds: alive until the flush
_fold_seen = set()            # parameters that already got a (possibly still queued) gradient in this backward pass
_fold_plain = set()           # parameters that ALSO take part in a forward through a plain (aten) path: never adopted unread
FOLD_POISON = os.environ.get("PTOPS_FOLD_POISON", "0") == "1"   # debug: queued outputs are NaN until the flush


class deferred_folds:
    """Context manager for a backward pass whose bias / LayerNorm-weight gradients are read only afterwards (or after a
    `fold_flush()`, as FlatDDP does before it packs a bucket): the ~115 fold launches become a few batched ones.
    Deferral is switched on for the CURRENT stream of `device` only (the C side keeps one queue per stream).

    What makes it safe, and what is checked:
      * a queued gradient must be adopted by autograd unread, i.e. `.grad` of every parameter must be None when the
        backward starts: pass `params` and the context refuses to defer (plain immediate folds) if any holds a value;
      * a parameter that gets a second gradient in the same pass - from a deferring producer (`fold_returning` sees the
        key twice) or from a plain aten path (`fold_mark_plain` at forward time) - is flushed before it is handed over;
      * `PTOPS_FOLD_POISON=1` fills every queued output with NaN until the flush: a premature read cannot stay silent."""

    def __init__(self, params=None, device=None):
        self.params = params
        self.device = device

    def __enter__(self):
        global _fold_deferred, _fold_stream
        _fold_seen.clear()
        if _test_backend is None and (self.params is None or all(p.grad is None for p in self.params)):
            dev = self.device
            if dev is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            handle = ctypes.c_void_p(_current_stream(dev).cuda_stream)
            load().ptops_fold_defer(handle, 2 if FOLD_POISON else 1)
            _fold_deferred, _fold_stream = True, (dev, handle)
        return self

    def __exit__(self, *exc):
        global _fold_deferred, _fold_stream
        if _fold_deferred:
            _fold_deferred = False
            load().ptops_fold_defer(_fold_stream[1], 0)
            fold_flush()
            _fold_stream = None
        _fold_plain.clear()
        return False


def fold_mark_plain(*keys):
    """Forward-time note: the parameters `keys` (data_ptr) are used through a plain autograd path in this forward, so
    their gradient may be accumulated by the engine; a deferring producer must not hand them a queued buffer."""
    _fold_plain.update(k for k in keys if k is not None)


def fold_returning(*keys):
    """Call where a backward node hands a possibly queued gradient of the parameter(s) `keys` (any hashables, e.g. the
    data_ptr of a leaf) to autograd.  The first gradient of a parameter is adopted unread; a SECOND one in the same
    backward pass (a module applied twice, or a plain aten path on the same parameter) is added to the first by the
    engine at once, so everything queued is folded first.  Keys must identify PARAMETERS: the address of a temporary
    can be recycled within the pass."""
    if _fold_deferred:
        if any(k in _fold_seen or k in _fold_plain for k in keys):
            fold_flush()
        _fold_seen.update(keys)


def _deferral_stream(dev):
    """The stream deferral is switched on for, as a torch stream object (the current stream outside `deferred_folds`)."""
    if _fold_stream is None:
        return torch.cuda.current_stream(dev)
    return torch.cuda.ExternalStream(_fold_stream[1].value or 0, device=dev)


def fold_flush():
    """Runs the queued weight gradients and folds on the stream they were queued on (no-op when nothing is queued).  A
    caller on ANOTHER stream (a backward node of a side branch inside a graph capture) is made to wait for that stream: what
    it asked for is visible to it when this returns."""
    wgrad_flush()
    dev = None
    if _fold_keepalive:
        dev = _fold_keepalive[0].device
        with _on(dev):
            handle = _fold_stream[1] if _fold_stream is not None else ctypes.c_void_p(_current_stream(dev).cuda_stream)
            _call("ptops_fold_flush", handle)
        _fold_keepalive.clear()
        for slot in _fold_arena.values():               # stream order: the next producers write after this flush
            slot[1] = 0
    if _fold_stream is not None:
        dev = _fold_stream[0]
        cur = torch.cuda.current_stream(dev)
        if cur.cuda_stream != (_fold_stream[1].value or 0):
            cur.wait_stream(_deferral_stream(dev))


_fold_arena = {}              # device index -> [uint8 arena tensor, bump offset]


def _fold_workspace(dev, stream, need, *outs):
    """The workspace of a column-sum producer: the shared per-stream scratch, or - under `deferred_folds` - a range
    of its own in a bump-allocated arena that is recycled at the flush (no tensor is created per call: host time
    matters for the launch-bound batch-16 models); -> a tensor or a raw device pointer, both accepted by `_ptr`."""
    if not _fold_deferred:
        return _scratch(dev, stream, need)
    need = (need + 255) & ~255
    slot = _fold_arena.get(dev.index)
    if slot is None or slot[1] + need > slot[0].numel():
        if slot is not None:
            _fold_keepalive.append(slot[0])             # still referenced by queued folds
            if _capture is not None:
                _capture.keepalive.append(slot[0])      # ... and by the captured launches, for the life of the graph
        slot = [torch.empty((max(256 << 20, 2 * need),), dtype=torch.uint8, device=dev), 0]
        _fold_arena[dev.index] = slot
    ptr = ctypes.c_void_p(slot[0].data_ptr() + slot[1])
    slot[1] += need
    for out in outs:
        # the output's storage must outlive the queued fold even if autograd drops the gradient, but the tensor object
        # must not gain a reference (AccumulateGrad only adopts a gradient it holds the sole reference to - otherwise
        # it would clone, i.e. READ, the not yet written values): hold the storage, not the tensor
        _fold_keepalive.append(out.untyped_storage())
    return ptr


def colsum(x):
    """x [M,N] float32 contiguous (N % 4 == 0) -> column sums [N] (deterministic)."""
    if _test_backend is not None:
        return x.sum(0)
    _check(x, "x", torch.float32, 2)
    M, N = x.shape
    with _on(x.device):
        stream = _current_stream(x.device)
        need = max(load().ptops_colsum_workspace_bytes(M, N), 16)
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
        ws = _fold_workspace(x.device, stream, need, out)
        _call("ptops_colsum", _ptr(x), M, N, _ptr(out), _ptr(ws), ctypes.c_void_p(stream.cuda_stream))
    return out


def max_tokens_supported(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.size(2) % 4 == 0 and x.size(1) >= 1 \
        and x.size(0) <= 65535


def max_tokens_forward(x):
    """x [B,M,C] float32 contiguous (C % 4 == 0) -> (max over tokens [B,C], arg-max token int32 [B,C])."""
    _check(x, "x", torch.float32, 3)
    B, M, C = x.shape
    with _on(x.device):
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, C), dtype=torch.int32, device=x.device)
        _call("ptops_max_tokens_forward", _ptr(x), B, M, C, _ptr(out), _ptr(arg), _stream(x))
    return out, arg


def max_tokens_backward(grad_out, arg, M):
    """grad_out [B,C], arg int32 [B,C] -> grad_x [B,M,C] (every element written)."""
    _check(grad_out, "grad_out", torch.float32, 2)
    _check(arg, "arg", torch.int32, 2)
    _same_device(grad_out, arg)
    B, C = grad_out.shape
    with _on(grad_out.device):
        gx = torch.empty((B, int(M), C), dtype=torch.float32, device=grad_out.device)
        _call("ptops_max_tokens_backward", _ptr(grad_out), _ptr(arg), B, int(M), C, _ptr(gx), _stream(grad_out))
    return gx


def gemm_split3(a, b, bias=None, a_k_contiguous=True, b_k_contiguous=True, out=None):
    """fp32 GEMM on the bf16 matrix pipe (csrc/gemm_split3.hip): out[M,N] = A . B^T (+ bias).

    a: [M,K] if a_k_contiguous else [K,M]; b: [N,K] if b_k_contiguous else [K,N]; both fp32 2-D with unit inner stride
    (row strides are passed on, so row-sliced views work).  K % 16 == 0."""
    for name, t in (("a", a), ("b", b)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise ValueError(f"gemm_split3: {name} must be a 2-D fp32 CUDA tensor with unit inner stride")
    _same_device(a, b)
    M, K = (a.shape[0], a.shape[1]) if a_k_contiguous else (a.shape[1], a.shape[0])
    N, K2 = (b.shape[0], b.shape[1]) if b_k_contiguous else (b.shape[1], b.shape[0])
    if K != K2:
        raise ValueError(f"gemm_split3: inner sizes differ ({K} vs {K2})")
    with _on(a.device):
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        if bias is not None:
            _check(bias, "bias", torch.float32, 1)
        _call("ptops_gemm_f32_split3", _ptr(a), a.stride(0), int(a_k_contiguous), _ptr(b), b.stride(0), int(b_k_contiguous),
              _ptr(out), out.stride(0), M, N, K, _ptr(bias) if bias is not None else None, _stream(a))
    return out


def gemm_split3_gelu(x, w, bias=None):
    """(x . w^T + bias, gelu of it) from one launch; x [M,K], w [N,K] fp32 with unit inner stride, K % 16 == 0."""
    for name, t in (("x", x), ("w", w)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise ValueError(f"gemm_split3_gelu: {name} must be a 2-D fp32 CUDA tensor with unit inner stride")
    _same_device(x, w)
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm_split3_gelu: inner sizes differ ({K} vs {w.shape[1]})")
    with _on(x.device):
        u = torch.empty((M, N), dtype=torch.float32, device=x.device)
        g = torch.empty((M, N), dtype=torch.float32, device=x.device)
        if bias is not None:
            _check(bias, "bias", torch.float32, 1)
        _call("ptops_gemm_f32_split3_gelu", _ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(u), u.stride(0), _ptr(g),
              g.stride(0), M, N, K, _ptr(bias) if bias is not None else None, _stream(x))
    return u, g


def gemm_split3_wgrad(dy, x, want_bias=False):
    """Weight (and bias) gradient of a dense layer on the bf16 matrix pipe: dy [M,N], x [M,K] (fp32, unit inner stride)
    -> (dy^T . x [N,K], column sums of dy [N] or None).  M % 16 == 0, K % 4 == 0 (and N % 4 == 0 with want_bias)."""
    for name, t in (("dy", dy), ("x", x)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise ValueError(f"gemm_split3_wgrad: {name} must be a 2-D fp32 CUDA tensor with unit inner stride")
    _same_device(dy, x)
    M, N = dy.shape
    if x.shape[0] != M:
        raise ValueError(f"gemm_split3_wgrad: row counts differ ({M} vs {x.shape[0]})")
    K = x.shape[1]
    with _on(dy.device):
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
        db = torch.empty((N,), dtype=torch.float32, device=dy.device) if want_bias else None
        need = int(load().ptops_gemm_f32_split3_wgrad_workspace(M, N, K))
        ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=dy.device)
        _call("ptops_gemm_f32_split3_wgrad", _ptr(dy), dy.stride(0), _ptr(x), x.stride(0), M, N, K, _ptr(dw), dw.stride(0),
              _ptr(db) if db is not None else None, _ptr(ws), need, _stream(dy))
    return dw, db


# ---- grouped weight gradients (include/ptops.h): queued during backward, two launches at the flush ------------------
GROUPED_WGRAD = os.environ.get("PTOPS_GROUPED_WGRAD", "1") != "0"
_wgrad_queue = []             # (dy, x, dW pointer, db pointer or 0, M, N, K): operands alive until the flush
_wgrad_plans = {}             # (device index, shapes) -> (host plan bytes, device plan, workspace bytes)
_wgrad_ws = {}                # device index -> workspace tensor
_wgrad_tables = {}            # plan key -> {bytes of a host table: the same table on the device} (the last few)
wgrad_table_uploads = [0]     # how often the device table had to be rewritten (diagnostics)


def wgrad_deferrable(dy, x):
    """True while `deferred_folds` is active: dy^T.x (and the column sums of dy) may be queued with `wgrad_defer`."""
    return (GROUPED_WGRAD and _fold_deferred and _test_backend is None and dy.is_cuda and dy.dtype == torch.float32
            and x.dtype == torch.float32 and dy.dim() == 2 and x.dim() == 2 and dy.stride(1) == 1 and x.stride(1) == 1
            and dy.size(0) == x.size(0) and dy.size(0) >= 16 and dy.size(0) % 16 == 0 and x.size(1) % 4 == 0
            and dy.device == _fold_stream[0] and _current_stream(dy.device).cuda_stream == (_fold_stream[1].value or 0))


def wgrad_defer(dy, x, want_bias):
    """Queues dW = dy^T.x [N,K] (and db = column sums of dy when `want_bias`, N % 4 == 0) for the grouped launch of the
    next `fold_flush()`; -> (dW, db) whose values exist only after that flush (same contract as a deferred fold: the
    caller hands them to autograd as the FIRST gradient of a leaf and announces that with `fold_returning`)."""
    M, N = dy.shape
    K = x.size(1)
    dev = dy.device
    dw = torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty((N,), dtype=torch.float32, device=dev) if want_bias else None
    if FOLD_POISON:
        dw.fill_(float("nan"))
        if db is not None:
            db.fill_(float("nan"))
    # the outputs' storage must outlive the queue even if autograd drops them; holding the tensor itself would make
    # AccumulateGrad clone (= read) instead of adopting it
    _wgrad_queue.append((dy, x, dw.data_ptr(), db.data_ptr() if db is not None else 0, M, N, K, dw.untyped_storage(),
                         db.untyped_storage() if db is not None else None))
    if _capture is not None and WGRAD_BRANCH_LAYERS > 0 and len(_wgrad_queue) >= WGRAD_BRANCH_LAYERS:
        wgrad_flush(join=False)        # inside a graph capture: a parallel branch beside the rest of backward (see wgrad_flush)
    return dw, db


# Inside a HIP-graph capture the queued weight gradients do not wait for the end of backward: every WGRAD_BRANCH_LAYERS
# layers the queue is launched on a side stream that forks from the backward stream at that point (two graph edges, not
# cross-stream events), so the grouped weight-gradient launches run BESIDE the input-gradient chain instead of after it -
# both families leave half of the matrix pipe idle on their own.  The branch joins where a result is needed
# (`fold_flush`: the end of backward, or a gradient of unknown provenance).  Never used outside a capture.
# MEASURED SLOWER and therefore off (0) by default - in-process-equivalent A/B of graphed steps on one box, 12 layers per
# launch vs one launch at the end: AdaPoinTr bs 32 31.5 vs 31.2 ms, image variant 24.4 vs 23.4, segmentation variant 33.8 vs
# 33.0.  Two matrix-pipe kernels side by side share L2 delivery and the LDS, and eight small grouped launches lose the one
# big launch's balance; the same verdict as the second-stream attempts of rounds 1-3, now without their event overhead.
WGRAD_BRANCH_LAYERS = int(os.environ.get("PTOPS_WGRAD_BRANCH_LAYERS", "0"))


def wgrad_flush(join=True):
    """Runs every queued weight gradient in one grouped launch (csrc/gemm_split3.hip) on the deferral stream - or, inside a
    graph capture with WGRAD_BRANCH_LAYERS > 0, on the capture's side stream; `join`: the deferral stream then waits for
    that branch (everything flushed so far is visible to it afterwards)."""
    branch = _capture is not None and WGRAD_BRANCH_LAYERS > 0
    if not _wgrad_queue:
        if branch and join:
            _capture.join_wgrad_branch()
        return
    import numpy as np
    queue = list(_wgrad_queue)
    _wgrad_queue.clear()
    dev = queue[0][0].device
    n = len(queue)
    lib = load()
    side = _capture.fork_wgrad_branch(dev, queue) if branch else None
    with _on(dev), (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        if side is not None:
            handle = ctypes.c_void_p(side.cuda_stream)
        else:
            handle = _fold_stream[1] if _fold_stream is not None else ctypes.c_void_p(_current_stream(dev).cuda_stream)
        key = (dev.index, tuple((q[4], q[5], q[6], q[3] != 0) for q in queue))
        plan = _wgrad_plans.get(key)
        if plan is None:
            mnk = np.ascontiguousarray([[q[4], q[5], q[6]] for q in queue], dtype=np.int32)
            flags = np.ascontiguousarray([1 if q[3] else 0 for q in queue], dtype=np.uint8)
            need = int(lib.ptops_wgrad_group_plan(mnk.ctypes.data, flags.ctypes.data, n, 0, None, 0))
            if need <= 0:
                raise PtopsError("wgrad_flush: unsupported problem in the queue")
            host = np.zeros(need, dtype=np.uint8)
            lib.ptops_wgrad_group_plan(mnk.ctypes.data, flags.ctypes.data, n, 0, host.ctypes.data, need)
            if _capture is not None:                          # a captured copy node: pinned source from the state's pool
                dev_plan = torch.empty(need, dtype=torch.uint8, device=dev)
                dev_plan.copy_(_capture.pinned_copy(host), non_blocking=True)
            else:
                dev_plan = torch.from_numpy(host).to(dev)
            plan = (host, dev_plan, int(lib.ptops_wgrad_group_workspace_bytes(host.ctypes.data)))
            if len(_wgrad_plans) > 16:
                _wgrad_plans.clear()
            _wgrad_plans[key] = plan
        host_plan, dev_plan, ws_bytes = plan
        ws = _wgrad_ws.get(dev.index)
        if ws is None or ws.numel() < ws_bytes:
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            _wgrad_ws[dev.index] = ws
        rows = np.zeros((n, 9), dtype=np.int64)                 # PtopsWgradProblem: 4 pointers, 3 strides, 4 int32
        assert lib.ptops_wgrad_problem_bytes() == 72
        for i, (dy, x, pw, pb, M, N, K, _sw, _sb) in enumerate(queue):
            rows[i] = (dy.data_ptr(), x.data_ptr(), pw, pb, dy.stride(0), x.stride(0), K, M | (N << 32), K)
        # The caching allocator hands a fixed-shape training loop the same blocks step after step, so the table of
        # pointers usually IS last step's table, still on the device: upload only when it differs (a host-to-device copy
        # in the middle of the stream costs ~0.2 ms of idle GPU on this stack: profiles/round3/idle_gaps_v1.log)
        # (the allocator may alternate between a few block assignments: keep the last few tables per plan)
        tables = _wgrad_tables.setdefault(key, {})
        if len(_wgrad_tables) > 16:
            _wgrad_tables.clear()
            tables = _wgrad_tables.setdefault(key, {})
        sig = rows.tobytes()
        rows_dev = tables.get(sig)
        if rows_dev is None:
            if len(tables) >= 8:                                # recycle the oldest device table (stream order: after
                rows_dev = tables.pop(next(iter(tables)))       # its last readers)
            else:
                rows_dev = torch.empty((n, 9), dtype=torch.int64, device=dev)
            if _capture is not None:                            # the copy node reads this table at every replay
                rows_dev.copy_(_capture.pinned_copy(rows).view(torch.int64).view(n, 9), non_blocking=True)
            else:
                rows_dev.copy_(torch.from_numpy(rows).pin_memory(), non_blocking=True)
            tables[sig] = rows_dev
            wgrad_table_uploads[0] += 1
        if _capture is not None:
            _capture.keepalive.append(plan)
        _call("ptops_gemm_f32_split3_wgrad_grouped", ctypes.c_void_p(rows.ctypes.data), _ptr(rows_dev), n,
              ctypes.c_void_p(host_plan.ctypes.data), _ptr(dev_plan), _ptr(ws), ws.numel(), handle)
        # `queue` (operands, output storages) dies here: the launches are queued behind it in stream order
    if branch and join:
        _capture.join_wgrad_branch()


def adamw_clip_step(rows_dev, n_tensors, chunks_dev, n_chunks, max_norm, beta1, beta2, eps, partial, norm_coef):
    """Gradient-norm clipping + AdamW over the device tables of utils/clip_adamw.py (csrc/optimizer.hip)."""
    with _on(rows_dev.device):
        _call("ptops_adamw_clip_step", _ptr(rows_dev), int(n_tensors), _ptr(chunks_dev), int(n_chunks), float(max_norm),
              float(beta1), float(beta2), float(eps), _ptr(partial), _ptr(norm_coef), _stream(rows_dev))


def _scratch(dev, stream, need):
    """One reusable scratch buffer per (device, stream), grown on demand (stream order makes reuse safe)."""
    key = (dev.index, stream.cuda_stream)
    ws = _colsum_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty((max(need, 1 << 22),), dtype=torch.uint8, device=dev)
        _colsum_ws[key] = ws
    return ws


def layer_norm_supported(x, weight, bias):
    return (x.is_cuda and x.dtype == torch.float32 and weight is not None and bias is not None
            and x.size(-1) % 4 == 0 and 4 <= x.size(-1) <= 1024 and weight.dtype == torch.float32)


def layer_norm_forward(x, weight, bias, eps):
    """x [rows, C] float32 contiguous -> (y [rows, C], mean [rows], rstd [rows])."""
    _check(x, "x", torch.float32, 2)
    _check(weight, "weight", torch.float32, 1)
    _check(bias, "bias", torch.float32, 1)
    dev = _same_device(x, weight, bias)
    rows, C = x.shape
    if weight.numel() != C or bias.numel() != C:
        raise ValueError("layer_norm_forward: weight / bias must have C elements")
    with _on(dev):
        y = torch.empty_like(x)
        mean = torch.empty((rows,), dtype=torch.float32, device=dev)
        rstd = torch.empty((rows,), dtype=torch.float32, device=dev)
        _call("ptops_layer_norm_forward", _ptr(x), _ptr(weight), _ptr(bias), rows, C, ctypes.c_float(float(eps)),
              _ptr(y), _ptr(mean), _ptr(rstd), _stream(x))
    return y, mean, rstd


def add_layer_norm_forward(x, res, weight, bias, eps):
    """x, res [rows, C] -> (s = x + res, y = LayerNorm(s), mean, rstd)."""
    _check(x, "x", torch.float32, 2)
    _check(res, "res", torch.float32, 2)
    _check(weight, "weight", torch.float32, 1)
    _check(bias, "bias", torch.float32, 1)
    dev = _same_device(x, res, weight, bias)
    rows, C = x.shape
    if tuple(res.shape) != (rows, C) or weight.numel() != C or bias.numel() != C:
        raise ValueError("add_layer_norm_forward: inconsistent shapes")
    with _on(dev):
        s = torch.empty_like(x)
        y = torch.empty_like(x)
        mean = torch.empty((rows,), dtype=torch.float32, device=dev)
        rstd = torch.empty((rows,), dtype=torch.float32, device=dev)
        _call("ptops_add_layer_norm_forward", _ptr(x), _ptr(res), _ptr(weight), _ptr(bias), rows, C,
              ctypes.c_float(float(eps)), _ptr(s), _ptr(y), _ptr(mean), _ptr(rstd), _stream(x))
    return s, y, mean, rstd


def layer_norm_backward(dy, x, mean, rstd, weight, extra=None):
    """-> (dx [rows, C] (+ extra), [3*C] = dgamma | dbeta | column sums of dx)."""
    _check(dy, "dy", torch.float32, 2)
    _check(x, "x", torch.float32, 2)
    _check(mean, "mean", torch.float32, 1)
    _check(rstd, "rstd", torch.float32, 1)
    _check(weight, "weight", torch.float32, 1)
    dev = _same_device(dy, x, mean, rstd, weight)
    rows, C = x.shape
    if tuple(dy.shape) != (rows, C) or mean.numel() != rows or rstd.numel() != rows or weight.numel() != C:
        raise ValueError("layer_norm_backward: inconsistent shapes")
    if extra is not None:
        _check(extra, "extra", torch.float32, 2)
        if tuple(extra.shape) != (rows, C) or extra.device != dev:
            raise ValueError("layer_norm_backward: extra must match x")
    with _on(dev):
        stream = _current_stream(dev)
        dx = torch.empty_like(x)
        dgb = torch.empty((3 * C,), dtype=torch.float32, device=dev)
        ws = _fold_workspace(dev, stream, max(load().ptops_layer_norm_backward_workspace_bytes(rows, C), 16), dgb)
        _call("ptops_layer_norm_backward", _ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(weight),
              _ptr(extra) if extra is not None else None, rows, C, _ptr(dx), _ptr(dgb), _ptr(ws),
              ctypes.c_void_p(stream.cuda_stream))
    return dx, dgb


def gelu_backward_colsum(x, dy):
    """x, dy [M,N] -> (dpre = dy * gelu'(x), column sums of dpre [N])."""
    _check(x, "x", torch.float32, 2)
    _check(dy, "dy", torch.float32, 2)
    dev = _same_device(x, dy)
    M, N = x.shape
    if tuple(dy.shape) != (M, N):
        raise ValueError("gelu_backward_colsum: shapes differ")
    with _on(dev):
        stream = _current_stream(dev)
        dpre = torch.empty_like(x)
        cs = torch.empty((N,), dtype=torch.float32, device=dev)
        ws = _fold_workspace(dev, stream, max(load().ptops_gelu_backward_colsum_workspace_bytes(M, N), 16), cs)
        _call("ptops_gelu_backward_colsum", _ptr(x), _ptr(dy), M, N, _ptr(dpre), _ptr(cs), _ptr(ws),
              ctypes.c_void_p(stream.cuda_stream))
    return dpre, cs


def batch_norm_rows_forward(x, weight, bias, eps, slope):
    """x [M,N] -> (y = LeakyReLU(batch-norm over rows), stats [2N] = {mean, rstd}, var [N] biased)."""
    _check(x, "x", torch.float32, 2)
    _check(weight, "weight", torch.float32, 1)
    _check(bias, "bias", torch.float32, 1)
    dev = _same_device(x, weight, bias)
    M, N = x.shape
    with _on(dev):
        stream = _current_stream(dev)
        ws = _scratch(dev, stream, max(load().ptops_batch_norm_rows_workspace_bytes(M, N), 16))
        y = torch.empty_like(x)
        stats = torch.empty((2 * N,), dtype=torch.float32, device=dev)
        var = torch.empty((N,), dtype=torch.float32, device=dev)
        _call("ptops_batch_norm_rows_forward", _ptr(x), _ptr(weight), _ptr(bias), M, N, ctypes.c_float(float(eps)),
              ctypes.c_float(float(slope)), _ptr(y), _ptr(stats), _ptr(var), _ptr(ws), ctypes.c_void_p(stream.cuda_stream))
    return y, stats, var


def batch_norm_rows_backward(dy, x, stats, weight, bias, slope):
    """-> (dx [M,N], dbeta_dgamma [2N])."""
    _check(dy, "dy", torch.float32, 2)
    _check(x, "x", torch.float32, 2)
    dev = _same_device(dy, x, stats, weight, bias)
    M, N = x.shape
    with _on(dev):
        stream = _current_stream(dev)
        ws = _scratch(dev, stream, max(load().ptops_batch_norm_rows_workspace_bytes(M, N), 16))
        dx = torch.empty_like(x)
        dbg = torch.empty((2 * N,), dtype=torch.float32, device=dev)
        _call("ptops_batch_norm_rows_backward", _ptr(dy), _ptr(x), _ptr(stats), _ptr(weight), _ptr(bias), M, N,
              ctypes.c_float(float(slope)), _ptr(dx), _ptr(dbg), _ptr(ws), ctypes.c_void_p(stream.cuda_stream))
    return dx, dbg


def edge_select(P, R, idx, gamma):
    """P [B,C,Nk], R [B,C,Nq], idx int32 [B,Nq,k], gamma float32 [C] (max over the neighbours where gamma >= 0, else min) ->
    (ysel [B,C,Nq], nsel int32 [B,C,Nq], ysum [B,C,Nq], partial [B,C,nblk,2])."""
    _check(P, "P", torch.float32, 3)
    _check(R, "R", torch.float32, 3)
    _check(idx, "idx", torch.int32, 3)
    _check(gamma, "gamma", torch.float32, 1)
    dev = _same_device(P, R, idx, gamma)
    B, C, Nk = P.shape
    Nq, k = idx.shape[1], idx.shape[2]
    if tuple(R.shape) != (B, C, Nq) or idx.shape[0] != B or gamma.numel() != C:
        raise ValueError("edge_select: inconsistent shapes")
    nblk = load().ptops_edge_select_blocks(Nq)
    with _on(dev):
        ysel = torch.empty((B, C, Nq), dtype=torch.float32, device=dev)
        nsel = torch.empty((B, C, Nq), dtype=torch.int32, device=dev)
        ysum = torch.empty((B, C, Nq), dtype=torch.float32, device=dev)
        partial = torch.empty((B, C, nblk, 2), dtype=torch.float32, device=dev)
        _call("ptops_edge_select", _ptr(P), _ptr(R), _ptr(idx), _ptr(gamma), B, C, Nk, Nq, k, _ptr(ysel),
              _ptr(nsel), _ptr(ysum), _ptr(partial), _stream(P))
    return ysel, nsel, ysum, partial


def edge_norm_forward(ysel, partial, gamma, beta, groups, k, eps, slope):
    """ysel [B,C,Nq], partial [B,C,nblk,2] (from edge_select) -> (out [B,C,Nq], mean_rstd [B,groups,2])."""
    _check(ysel, "ysel", torch.float32, 3)
    _check(partial, "partial", torch.float32, 4)
    _check(gamma, "gamma", torch.float32, 1)
    _check(beta, "beta", torch.float32, 1)
    dev = _same_device(ysel, partial, gamma, beta)
    B, C, Nq = ysel.shape
    with _on(dev):
        out = torch.empty_like(ysel)
        mean_rstd = torch.empty((B, groups, 2), dtype=torch.float32, device=dev)
        _call("ptops_edge_norm_forward", _ptr(ysel), _ptr(partial), _ptr(gamma), _ptr(beta), B, C, Nq, partial.size(2),
              int(groups), int(k), ctypes.c_float(float(eps)), ctypes.c_float(float(slope)), _ptr(out), _ptr(mean_rstd),
              _stream(ysel))
    return out, mean_rstd


def edge_norm_backward(grad_out, ysel, ysum, P, R, idx, nsel, gamma, beta, mean_rstd, slope):
    """-> (dP [B,C,Nk], dR [B,C,Nq], sums [B,C,2] = per (b,c) {sum dz, sum dz*yhat})."""
    for t, name in ((grad_out, "grad_out"), (ysel, "ysel"), (ysum, "ysum"), (P, "P"), (R, "R")):
        _check(t, name, torch.float32, 3)
    _check(idx, "idx", torch.int32, 3)
    _check(nsel, "nsel", torch.int32, 3)
    dev = _same_device(grad_out, ysel, ysum, P, R, idx, nsel, gamma, beta, mean_rstd)
    B, C, Nq = ysel.shape
    Nk, k, groups = P.size(2), idx.size(2), mean_rstd.size(1)
    with _on(dev):
        stream = _current_stream(dev)
        dP = torch.empty_like(P)
        dR = torch.empty_like(R)
        sums = torch.empty((B, C, 2), dtype=torch.float32, device=dev)
        ws = torch.empty((load().ptops_edge_norm_backward_workspace_bytes(B, C, Nk, Nq, k, groups) // 4 + 4,),
                         dtype=torch.float32, device=dev)
        _call("ptops_edge_norm_backward", _ptr(grad_out), _ptr(ysel), _ptr(ysum), _ptr(P), _ptr(R), _ptr(idx), _ptr(nsel),
              _ptr(gamma), _ptr(beta), _ptr(mean_rstd), B, C, Nk, Nq, k, groups, ctypes.c_float(float(slope)), _ptr(dP),
              _ptr(dR), _ptr(sums), _ptr(ws), ctypes.c_void_p(stream.cuda_stream))
    return dP, dR, sums


def edge_scatter(dense, sel, idx, nsel, Nk):
    """dense/sel [B,C,Nq], idx int32 [B,Nq,k], nsel int32 [B,C,Nq] -> T [B,C,Nk]."""
    _check(dense, "dense", torch.float32, 3)
    _check(sel, "sel", torch.float32, 3)
    _check(idx, "idx", torch.int32, 3)
    _check(nsel, "nsel", torch.int32, 3)
    dev = _same_device(dense, sel, idx, nsel)
    B, C, Nq = dense.shape
    k = idx.shape[2]
    with _on(dev):
        T = torch.empty((B, C, int(Nk)), dtype=torch.float32, device=dev)
        _call("ptops_edge_scatter", _ptr(dense), _ptr(sel), _ptr(idx), _ptr(nsel), B, C, int(Nk), Nq, k, _ptr(T),
              _stream(dense))
    return T


def set_option(name, value):
    """A/B switch of the library (include/ptops.h, ptops_set_option); -> the previous value."""
    old = load().ptops_set_option(name.encode(), int(value))
    if old < 0:
        raise PtopsError(f"unknown libptops option {name!r}")
    return old


def get_option(name):
    return load().ptops_get_option(name.encode())
