#!/usr/bin/env python
"""Headline benchmark: clouds/sec of one full AdaPoinTr train step on PCN-shaped synthetic data.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + get_loss (denoise kNN + 3 Chamfer) + backward + gradient averaging (RCCL) +
clip_grad_norm_(10) + AdamW step on one batch of 32 clouds per GPU (2048-point partial ->
16384-point completion; BASELINE.json configs[1]), fp32 throughout, inputs resident in HBM.
Rank 0 prints ONE JSON line.  It carries
  roofline     — the Chamfer 16384x16384 nearest-neighbour launch (north_star's lead primitive): average launch
                 duration measured with events on the launching stream during the timed steps; when the
                 bf16x3 MFMA filter runs it, executed MFMA FLOPs against the dense bf16 peak (bound "mfma"),
                 otherwise algorithmic FLOPs against the fp32 vector peak; plus the same launch priced in
                 algorithmic FLOPs and algorithmic HBM bytes;
  cpu_baseline — the same train step run on the host cores through the CPU oracle ("port": the
                 reference has no CPU implementation of Chamfer / FPS), on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: see imgadapointr_amd/__init__.py

import torch  # noqa: E402
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from imgadapointr_amd import _native  # noqa: E402
from imgadapointr_amd.models import build_model_from_cfg  # noqa: E402
from imgadapointr_amd.models import Transformer_utils as TU  # noqa: E402
from imgadapointr_amd.tools import builder  # noqa: E402
from imgadapointr_amd.tools.runner import TrainStep  # noqa: E402
from imgadapointr_amd.utils.config import cfg_from_yaml_file  # noqa: E402
from imgadapointr_amd.utils.flat_ddp import FlatDDP  # noqa: E402
from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm  # noqa: E402
from imgadapointr_amd.utils.graph_runner import GraphedTrainStep  # noqa: E402

PEAK_FP32_VECTOR_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA peak (same table; 2:1 sparsity figures are not used)
PEAK_HBM_GBS = 8000.0
FLOP_PER_PAIR = 8                 # 3 sub + 3 mul + 2 add per squared distance
N_PARTIAL, N_GT, BATCH = 2048, 16384, 32


# --model: the three model families of BASELINE.json's configs.  The headline metric (configs[1] / [3]) is "adapointr";
# "img" = configs[2] (image cross-attention fusion, bs 16), "seg" = configs[4] (image + segmentation branch, bs 16).
MODELS = {
    "adapointr": dict(yaml="cfgs/PCN_models/AdaPoinTr.yaml", batch=32, img=False, cls=False, config="configs[1]",
                      name="AdaPoinTr"),
    "img": dict(yaml="cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml", batch=16, img=True, cls=False,
                config="configs[2]", name="ImgResNetEncAdaPoinTrVariableLoss (ImgAdaPoinTr)"),
    "seg": dict(yaml="cfgs/SegImgPCN_models/ImgEncSegDecAdaPoinTrVariableLoss.yaml", batch=16, img=True, cls=True,
                config="configs[4]", name="ImgEncSegDecAdaPoinTrVariableLoss (ImgEncSegDecAPTr)"),
}


def load_cfg(which="adapointr"):
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        return cfg_from_yaml_file(MODELS[which]["yaml"])
    finally:
        os.chdir(cwd)


class ChamferProbe:
    """Times every Chamfer forward launch of a given shape with events on the launching stream."""

    def __init__(self, n, m):
        self.shape = (n, m)
        self.events = []
        self.orig = _native.chamfer_forward
        self.enabled = False

    def __call__(self, xyz1, xyz2):
        if self.enabled and (xyz1.shape[1], xyz2.shape[1]) == self.shape:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()          # current stream == the stream libptops launches on
            out = self.orig(xyz1, xyz2)
            e.record()
            self.events.append((s, e, xyz1.shape[0]))
            return out
        return self.orig(xyz1, xyz2)

    def summary(self):
        if not self.events:
            return None
        ms = [s.elapsed_time(e) for s, e, _ in self.events]
        B = self.events[0][2]
        avg_ms = sum(ms) / len(ms)
        n, m = self.shape
        pairs = 2.0 * B * n * m
        alg_flops = FLOP_PER_PAIR * pairs
        alg_bytes = 20.0 * B * (n + m)
        algos = _native.chamfer_algorithm_log.get((B, n, m), (0, 0))
        names = {0: "auto", 1: "brute-force chamfer_nn_kernel (VALU)", 2: "pruned sweep (VALU)",
                 3: "bf16x3 MFMA filter + exact re-evaluation (chamfer_mfma_kernel)"}
        on_mfma = all(a == _native.CHAMFER_MFMA for a in algos)
        # HBM traffic and pipe activity of this launch come from separate rocprofv3 --pmc passes of the same kernel
        # (cannot be collected from inside the timed run); their summary is committed.
        pmc_name = "chamfer_mfma_pmc.json" if on_mfma else "chamfer_pmc.json"
        pmc_round = "round4" if on_mfma else "round1"      # the MFMA search was re-profiled on the round-4 kernel
        pmc_file = os.path.join(ROOT, "profiles", pmc_round, pmc_name)
        traffic, pmc = None, {}
        if os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                pmc = json.load(fh).get("derived", {})
            traffic = pmc.get("hbm_traffic_bytes_corrected")
        out = {
            "kernel": "Chamfer 16384x16384 forward launch (exact nearest neighbours, both directions; per direction the "
                      "adaptive policy ran: " + " / ".join(names[a] for a in algos) + ")",
            "launches": len(ms), "avg_launch_ms": round(avg_ms, 4), "traffic": traffic,
            "traffic_source": f"profiles/{pmc_round}/{pmc_name} (FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)",
            "algorithmic_flops_per_launch": alg_flops, "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic": {"achieved": round(alg_flops / (avg_ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                            "note": "8 FLOP per pair (3 sub, 3 mul, 2 add), every pair of both directions"},
            "hbm": {"achieved": round(alg_bytes / (avg_ms * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(alg_bytes / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)},
        }
        if on_mfma:
            # two v_mfma_f32_32x32x16_bf16 per 32 x 32 pairs = 64 executed FLOP per pair (21 of the 32 k-slots carry
            # piece products); the matrix pipe is the binding unit, the exact phase and the minima ride on the VALU
            executed = 64.0 * pairs
            achieved = executed / (avg_ms * 1e-3) / 1e12
            out.update({"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_BF16_MFMA_TFLOPS, 4), "executed_flops_per_launch": executed,
                        "mfma_busy_fraction_pmc": pmc.get("mfma_busy_fraction"),
                        "clock_ghz_pmc": pmc.get("clock_ghz")})
        else:
            achieved = alg_flops / (avg_ms * 1e-3) / 1e12
            out.update({"bound": "valu-fp32", "achieved": round(achieved, 3), "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 4),
                        "valu_busy_fraction_pmc": pmc.get("valu_busy_fraction")})
        return out


class AttentionProbe:
    """Events around every attention forward / backward launch of the library (the second family of MFMA-bound
    hand-written kernels).  From 128 tokens up the kernels run on the bf16 matrix pipe in three-piece fp32-accurate
    arithmetic (csrc/attention.hip: attn_fwd_split3_kernel, attn_bwd_dq_split3_kernel, attn_bwd_dkv_split3_kernel):
    every fp32 product FLOP is executed as 6 bf16 piece-product FLOPs, priced against the 2.5 PFLOP/s dense bf16 peak."""

    def __init__(self):
        self.fwd, self.bwd = _native.attention_forward, _native.attention_backward
        self.enabled = False
        self.rec = {"forward": [], "backward": []}

    @staticmethod
    def _pairs(q, k, split_q, limit_k):
        B, H, Nq, _ = q.shape
        Nk = k.size(2)
        return B * H * ((split_q * limit_k + (Nq - split_q) * Nk) if (split_q > 0 and limit_k < Nk) else Nq * Nk)

    @staticmethod
    def _on_bf16_pipe(q, k, backward, fused):
        Nq, Nk = q.size(2), k.size(2)
        if backward:
            return (not fused) and Nq >= 128 and Nk >= 128 and not _native.get_option("PTOPS_ATTN_BWD_F32")
        return Nq >= 128 and not _native.get_option("PTOPS_ATTN_FWD_F32")

    def forward(self, q, k, v, split_q, limit_k, scale):
        if not self.enabled:
            return self.fwd(q, k, v, split_q, limit_k, scale)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = self.fwd(q, k, v, split_q, limit_k, scale)
        e.record()
        self.rec["forward"].append((s, e, 4.0 * 64 * self._pairs(q, k, split_q, limit_k),         # 2 GEMMs
                                    self._on_bf16_pipe(q, k, False, False)))
        return out

    def backward(self, q, k, v, out, dout, lse, split_q, limit_k, scale, dq, dk, dv, fused=True):
        if not self.enabled:
            return self.bwd(q, k, v, out, dout, lse, split_q, limit_k, scale, dq, dk, dv, fused=fused)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = self.bwd(q, k, v, out, dout, lse, split_q, limit_k, scale, dq, dk, dv, fused=fused)
        e.record()
        gemms = 5.0 if fused else 7.0       # the dq + dk/dv pair evaluates S and dP twice
        self.rec["backward"].append((s, e, 2.0 * gemms * 64 * self._pairs(q, k, split_q, limit_k),
                                     self._on_bf16_pipe(q, k, True, fused)))
        return r

    def summary(self, steps):
        out = {"bound": "mfma", "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
               "note": "achieved = fp32 product FLOPs of the launches (forward 2 GEMMs, dq + dk/dv pair 7, single-pass "
                       "backward 5) per second; executed = x 6 for launches on the bf16 matrix pipe (three-piece split, six "
                       "piece products per fp32 product), x 1 for the f32-MFMA kernels; frac = executed / dense bf16 peak. "
                       "pmc = separate rocprofv3 --pmc passes of the same kernels as they are now (profiles/round4/attention_pmc.json)"}
        pmc_file = os.path.join(ROOT, "profiles", "round4", "attention_pmc.json")
        if os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                out["pmc"] = [{"kernel": k["kernel"], "workgroups": k["grid_threads"] // k["workgroup"],
                               "launch_us": k["avg_duration_us_under_pmc"], "mfma_busy": k.get("mfma_utilisation"),
                               "traffic": k["counters"].get("hbm_traffic_bytes")} for k in json.load(fh)["kernels"]]
        for kind, recs in self.rec.items():
            if not recs:
                continue
            ms = sum(s.elapsed_time(e) for s, e, _, _ in recs)
            fl = sum(f for _, _, f, _ in recs)
            ex = sum(f * (6.0 if bf else 1.0) for _, _, f, bf in recs)
            out[kind] = {"launches_per_step": len(recs) / steps, "ms_per_step": round(ms / steps, 3),
                         "launches_on_bf16_pipe": sum(1 for r in recs if r[3]) / steps,
                         "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "executed": round(ex / (ms * 1e-3) / 1e12, 1),
                         "frac": round(ex / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}
        return out


class GemmProbe:
    """Events around every dense-layer GEMM that runs on libptops' split-bf16 kernels during one timed step (forward,
    input gradient, weight gradients - per layer or the grouped launch of all queued layers incl. its reduction -,
    Linear + GELU): fp32 product FLOPs per second (`achieved`) and
    the executed bf16 piece-product FLOPs (x 6) against the dense bf16 peak (`frac`).  The remaining GEMMs of the step are
    hipBLASLt / rocBLAS fp32-MFMA kernels (profiles/round2/step_breakdown_v4.txt)."""
    NAMES = ("gemm_split3", "gemm_split3_wgrad", "gemm_split3_gelu")

    def __init__(self):
        self.orig = {n: getattr(_native, n) for n in self.NAMES}
        self.enabled = False
        self.rec = []
        for n in self.NAMES:
            setattr(_native, n, self._wrap(n))
        # the grouped weight-gradient launch (all queued layers in one kernel + one reduction): its FLOPs are the queue's
        self.orig_flush = _native.wgrad_flush

        def flush(*a, **kw):
            if not (self.enabled and _native._wgrad_queue):
                return self.orig_flush(*a, **kw)
            flops = sum(2.0 * q[4] * q[5] * q[6] for q in _native._wgrad_queue)
            n = len(_native._wgrad_queue)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self.orig_flush(*a, **kw)
            e.record()
            self.rec.append(("gemm_split3_wgrad_grouped", s, e, flops))
            self.grouped_problems = n
        _native.wgrad_flush = flush

    def _wrap(self, name):
        fn = self.orig[name]

        def call(*args, **kw):
            if not self.enabled:
                return fn(*args, **kw)
            a, b = args[0], args[1]
            if name == "gemm_split3_wgrad":                       # dy [M,N], x [M,K] -> [N,K]
                flops = 2.0 * a.shape[0] * a.shape[1] * b.shape[1]
            else:
                ak = kw.get("a_k_contiguous", args[3] if len(args) > 3 else True) if name == "gemm_split3" else True
                bk = kw.get("b_k_contiguous", args[4] if len(args) > 4 else True) if name == "gemm_split3" else True
                M, K = (a.shape[0], a.shape[1]) if ak else (a.shape[1], a.shape[0])
                N = b.shape[0] if bk else b.shape[1]
                flops = 2.0 * M * N * K
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*args, **kw)
            e.record()
            self.rec.append((name, s, e, flops))
            return out
        return call

    def restore(self):
        for n in self.NAMES:
            setattr(_native, n, self.orig[n])
        _native.wgrad_flush = self.orig_flush

    def summary(self):
        out = {"bound": "mfma", "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
               "note": "dense-layer GEMMs on the split-bf16 kernels in one timed step; achieved = fp32 product FLOP/s, "
                       "executed = x 6 (six bf16 piece products per fp32 product), frac = executed / dense bf16 peak"}
        by = {}
        for name, s, e, fl in self.rec:
            d = by.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e)
            d[2] += fl
        tot_ms = sum(d[1] for d in by.values())
        tot_fl = sum(d[2] for d in by.values())
        for name, (n, ms, fl) in by.items():
            out[name] = {"launches_per_step": n, "ms_per_step": round(ms, 3), "achieved": round(fl / (ms * 1e-3) / 1e12, 1)}
        if "gemm_split3_wgrad_grouped" in out:
            out["gemm_split3_wgrad_grouped"]["layers_per_launch"] = getattr(self, "grouped_problems", None)
            pmc = os.path.join(ROOT, "profiles", "round3", "wgrad_grouped_pmc.json")   # separate rocprofv3 --pmc passes
            if os.path.exists(pmc):
                with open(pmc) as fh:
                    k = json.load(fh)["kernels"][0]
                d = k.get("derived", {})
                out["gemm_split3_wgrad_grouped"]["pmc"] = {
                    "source": "profiles/round3/wgrad_grouped_pmc.json", "launch_us": k.get("avg_duration_us_under_pmc"),
                    "mfma_busy": d.get("mfma_busy_fraction"), "valu_busy": d.get("valu_busy_fraction"),
                    "waves_per_simd": d.get("waves_resident_per_simd"), "clock_ghz": d.get("clock_ghz"),
                    "traffic": d.get("hbm_traffic_bytes"), "fetch_bytes_raw": d.get("fetch_bytes_raw")}
        if tot_ms > 0:
            out.update({"ms_per_step": round(tot_ms, 3), "achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                        "executed": round(6 * tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                        "frac": round(6 * tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)})
        return out


class PrimitiveProbe:
    """Events around every FPS and kNN launch of one step (algorithmic bytes 12 N + 4 npoint resp.
    12 (N + S) + 4 S k per cloud; work = (npoint - 1) N point updates resp. S N pair evaluations).  Both are
    latency / selection bound with the clouds resident on chip, so the HBM fraction is reported for what it is."""

    def __init__(self):
        self.fps, self.knn = _native.furthest_point_sample, _native.knn
        self.enabled = False
        self.rec = {"fps": [], "knn": []}

    def _timed(self, kind, fn, args, nbytes, work):
        if not self.enabled:
            return fn(*args)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*args)
        e.record()
        self.rec[kind].append((s, e, nbytes, work))
        return out

    def furthest_point_sample(self, xyz, npoint):
        B, N, _ = xyz.shape
        return self._timed("fps", self.fps, (xyz, npoint), B * (12.0 * N + 4.0 * npoint), B * (npoint - 1.0) * N)

    def knn_call(self, k, xyz, new_xyz):
        B, N, _ = xyz.shape
        S = new_xyz.shape[1]
        return self._timed("knn", self.knn, (k, xyz, new_xyz), B * (12.0 * (N + S) + 4.0 * S * k), B * 1.0 * S * N)

    def summary(self):
        out = {}
        for kind, unit in (("fps", "Gupdate/s"), ("knn", "Gpair/s")):
            recs = self.rec[kind]
            if not recs:
                continue
            # HBM traffic and pipe activity: separate rocprofv3 --pmc passes over the same launches
            # (benchmarks/primitives_only.py, benchmarks/pmc_passes.sh), averages per launch
            pmc_dir = "round4"                                         # both were re-profiled on the round-4 kernels
            pmc_file = os.path.join(ROOT, "profiles", pmc_dir, f"{kind}_pmc.json")
            pmc = {}
            if os.path.exists(pmc_file):
                with open(pmc_file) as fh:
                    ks = json.load(fh).get("kernels", [])
                pmc = ks[0].get("derived", {}) if ks else {}
            ms = sum(s.elapsed_time(e) for s, e, _, _ in recs)
            nbytes, work = sum(r[2] for r in recs), sum(r[3] for r in recs)
            gbs = nbytes / (ms * 1e-3) / 1e9
            out[kind] = {"launches_per_step": len(recs), "ms_per_step": round(ms, 3),
                         "work_rate": round(work / (ms * 1e-3) / 1e9, 1), "work_unit": unit,
                         "hbm": {"achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(gbs / PEAK_HBM_GBS, 5), "algorithmic_bytes_per_step": nbytes},
                         "traffic": (round(pmc["hbm_traffic_bytes"] * len(recs)) if "hbm_traffic_bytes" in pmc else None),
                         "traffic_source": f"profiles/{pmc_dir}/{kind}_pmc.json (per-launch average x launches; FETCH_SIZE x 2 "
                                           "upper bound + WRITE_SIZE)",
                         "pmc": {k: round(pmc[k], 4) for k in ("valu_busy_fraction", "salu_busy_fraction",
                                                               "waves_resident_per_simd", "sq_wait_any_share_of_wave_cycles",
                                                               "lds_issue_busy_fraction") if k in pmc}}
        return out


def cpu_baseline(cfg, spec, budget_s=20.0):
    """The same train step on the host cores through the CPU oracle, B=2 (BASELINE configs[0])."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import ref as oracle
    from oracle_backend import OracleBackend
    oracle.build()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)  # the GPU box's CPU share for one GPU; more threads only oversubscribe
    torch.set_num_threads(cores)
    oracle.set_num_threads(cores)
    _native.set_backend_for_testing(OracleBackend())
    try:
        torch.manual_seed(0)
        model = build_model_from_cfg(cfg.model)
        if spec["cls"]:
            for p in model.base_model.segmentator.parameters():
                p.requires_grad = False
        model.train()
        opt = builder.build_optimizer(model, cfg)
        step = TrainStep(model, opt)
        B = 2
        partial = torch.rand(B, N_PARTIAL, 3) - 0.5
        gt = torch.rand(B, N_GT, 3) - 0.5
        extra = ((torch.rand(B, 3, 224, 224),) if spec["img"] else ()) + \
                ((torch.eye(16)[:B].contiguous(),) if spec["cls"] else ())
        _step = step
        step = lambda p, g: _step(p, g, extra)        # noqa: E731
        tw = time.time()
        step(partial, gt)  # warm-up
        tw = time.time() - tw
        max_steps = max(1, min(12, int(budget_s / max(tw, 1e-3))))
        t0 = time.time()
        n = 0
        while n < max_steps:
            step(partial, gt)
            n += 1
            if time.time() - t0 > budget_s:
                break
        dt = time.time() - t0
        # the same step at the metric's batch size (one timed step after one warm-up) and the
        # primitives alone at the hot-path shapes (per cloud, the oracle's OpenMP loops over `cores` threads)
        extra_rows = {}
        if spec["config"] == "configs[1]":
            import numpy as np
            B32 = BATCH
            p32, g32 = torch.rand(B32, N_PARTIAL, 3) - 0.5, torch.rand(B32, N_GT, 3) - 0.5
            step(p32, g32)
            t1 = time.time()
            step(p32, g32)
            d32 = time.time() - t1
            extra_rows["bs32"] = {"value": round(B32 / d32, 3), "unit": "clouds/s", "ms_per_step": round(d32 * 1e3, 1),
                                  "sample": "1 full train step at bs=32 after 1 warm-up step"}
            rng = np.random.default_rng(0)
            ops = {}

            def t_op(fn, reps):
                fn()
                t = time.time()
                for _ in range(reps):
                    fn()
                return (time.time() - t) / reps * 1e3

            Bo = 4
            for n_, m_ in ((2048, 2048), (512, 16384), (16384, 16384)):
                a_, b_ = rng.random((Bo, n_, 3), dtype=np.float32) - 0.5, rng.random((Bo, m_, 3), dtype=np.float32) - 0.5
                ops[f"chamfer_forward {n_}x{m_}"] = round(t_op(lambda: oracle.chamfer_forward(a_, b_), 2) / Bo, 3)
            for S_, N_, k_ in ((2048, 2048, 16), (512, 2048, 16), (512, 512, 16), (256, 512, 16), (256, 256, 8),
                               (576, 256, 8), (512, 512, 8), (64, 16384, 32)):
                c_, q_ = rng.random((Bo, N_, 3), dtype=np.float32) - 0.5, rng.random((Bo, S_, 3), dtype=np.float32) - 0.5
                ops[f"knn {S_}x{N_} k{k_}"] = round(t_op(lambda: oracle.knn(k_, c_, q_), 3) / Bo, 3)
            for N_, np_ in ((2048, 512), (512, 256), (2048, 256), (2048, 64)):
                c_ = rng.random((Bo, N_, 3), dtype=np.float32) - 0.5
                ops[f"fps {N_}->{np_}"] = round(t_op(lambda: oracle.furthest_point_sample(c_, np_), 3) / Bo, 3)
            extra_rows["ops_ms_per_cloud"] = ops
    finally:
        _native.set_backend_for_testing(None)
    return {**extra_rows, "value": round(B * n / dt, 3), "unit": "clouds/s", "cores": cores, "kind": "port",
            "sample": f"{n} full train steps (fwd+loss+bwd+clip+AdamW) at bs={B}, 2048->16384, eager fp32 torch on "
                      f"{cores} threads, Chamfer/FPS/kNN/gathers through the CPU oracle (OpenMP)",
            "ms_per_step": round(dt / n * 1e3, 1)}


def make_workload(which, device, batch=None, views=1, seed=0, wrap=None):
    """Model (random init), production TrainStep and one synthetic batch of a model family, resident on `device`."""
    spec = MODELS[which]
    B = batch or spec["batch"]
    cfg = load_cfg(which)
    torch.manual_seed(0)                     # identical weights on every rank
    model = build_model_from_cfg(cfg.model).to(device)
    if spec["cls"]:                          # the runner freezes the segmentator (tools/runner.py:55-66)
        for p in model.base_model.segmentator.parameters():
            p.requires_grad = False
    model.train()
    wrapped = wrap(model) if wrap is not None else model
    optimizer = builder.build_optimizer(wrapped, cfg)
    step = TrainStep(wrapped, optimizer, getattr(cfg, "grad_norm_clip", 10), 1)
    torch.manual_seed(seed)                  # per-rank data, main.py:57 of the reference
    partial = (torch.rand(B, N_PARTIAL, 3) - 0.5).to(device)
    gt = (torch.rand(B, N_GT, 3) - 0.5).to(device)
    extra = []
    if spec["img"]:
        shape = (B, 3, 224, 224) if views == 1 else (B, views, 3, 224, 224)
        extra.append(torch.rand(*shape).to(device))
    if spec["cls"]:
        cls_vec = torch.zeros(B, 16)
        cls_vec[torch.arange(B), torch.tensor([0, 4, 8, 15])[torch.arange(B) % 4]] = 1.0    # SegImgPCNDataset.py:5-12
        extra.append(cls_vec.to(device))
    return model, cfg, spec, step, (partial, gt, tuple(extra)), B


def time_steps(step, batch, steps, warmup):
    partial, gt, extra = batch
    for _ in range(warmup):
        step(partial, gt, extra)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step(partial, gt, extra)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, last


def run_variant(which, device, steps=10, warmup=5, note=lambda m: None):
    """One of BASELINE's other single-GPU configurations, timed after the headline region so that the driver's JSON
    carries it: `steps` eager train steps, then the same step replayed from a HIP graph (GraphedTrainStep: 3 eager steps
    in its own workspaces, capture, a capture-time check against the eager step, then `steps` timed replays).  `value` is
    the graphed rate when the graph passed its check, else the eager rate."""
    model, cfg, spec, step, batch, B = make_workload(which, device)
    eager_s, _ = time_steps(step, batch, steps, warmup)
    note(f"variant {which}: eager {eager_s * 1e3:.2f} ms/step")
    gstep = GraphedTrainStep(step, warmup=3, verify=True)
    graph_s, last = time_steps(gstep, batch, steps, 3 + 1 + 2)
    graphed = gstep.enabled and gstep.replays >= steps
    note(f"variant {which}: graphed {graph_s * 1e3:.2f} ms/step (enabled={gstep.enabled}, capture check {gstep.verified})")
    best = graph_s if graphed else eager_s
    out = {"metric": f"clouds/sec (train step, bs={B}/GPU), {spec['name']} PCN 2048->16384", "value": round(B / best, 3),
           "unit": "clouds/s", "ms_per_step": round(best * 1e3, 3), "steps": steps, "n_gpus": 1, "dtype": "f32",
           "data": "synthetic", "graphed": graphed, "eager_ms_per_step": round(eager_s * 1e3, 3),
           "graphed_ms_per_step": round(graph_s * 1e3, 3) if graphed else None,
           "graph_capture_check": gstep.verified, "graph_disabled_reason": gstep.disabled_reason,
           "last_losses_x1000": [round(float(x) * 1000, 4) for x in last],
           "config": {"workload": f"{spec['name']} {spec['yaml']}, PCN 2048->16384, bs={B}/GPU"
                                  + (", one 3x224x224 render per sample" if spec["img"] else "")
                                  + (", 16-class one-hot + frozen GDANet segmentator" if spec["cls"] else "")
                                  + f", full train step (BASELINE {spec['config']})"}}
    del gstep, step, model
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", choices=sorted(MODELS), default="adapointr",
                    help="adapointr = BASELINE configs[1] (the headline metric); img = configs[2]; seg = configs[4]")
    ap.add_argument("--views", type=int, default=1,
                    help="img / seg only: renders per object in the batch ([B,V,3,224,224]; the README's 6 / 12-view "
                         "protocol).  1 = the reference's train step (one render per sample).  With V > 1 a step "
                         "trains on B objects x V views = B*V samples; the point branch runs once per object")
    ap.add_argument("--batch", type=int, default=None,
                    help="clouds (objects) per GPU; default 32 for adapointr, 16 for img / seg (BASELINE.json configs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the train step from a HIP graph (GraphedTrainStep; one process, no data-parallel wrapper); "
                         "auto = on for --model img | seg in one process, off for the device-bound headline step and under "
                         "torch.distributed.  The per-kernel roofline probes are "
                         "Python-side and see nothing of a replay: the FIRST timed step runs eagerly and carries them")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the `variants` block (configs[2] / [4] at bs 16, eager and graphed, 10 timed steps each) that "
                         "the default one-GPU run appends after the headline region")
    ap.add_argument("--no-gemm-tuning", action="store_true", help="leave the GEMM kernel choice to the library default")
    args = ap.parse_args()

    t_start = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # PTOPS_BENCH_REHEARSAL=1: all ranks share GPU 0 and talk over gloo — lets the multi-rank code path
    # (FlatDDP hooks, barriers, max-over-ranks timing) be rehearsed on a one-GPU box.  Never a measurement.
    rehearsal = os.environ.get("PTOPS_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    force_ddp = os.environ.get("PTOPS_BENCH_FORCE_DDP") == "1"
    # PTOPS_BENCH_FORCE_DDP=1 under torch.distributed.run with ONE rank: the real RCCL call sequence (broadcasts, async
    # AVG all-reduces from the gradient hooks, waits) on a one-GPU box.  Never the default, never a scaling number.
    single_rank_rccl = force_ddp and world == 1 and "MASTER_ADDR" in os.environ and not rehearsal
    if world > 1 or single_rank_rccl:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
    _native.load()
    gemm = "library default" if args.no_gemm_tuning else enable_tuned_gemm()

    spec = MODELS[args.model]
    if args.views != 1 and not spec["img"]:
        raise SystemExit("--views needs a model with an image branch (--model img | seg)")
    # PTOPS_BENCH_FORCE_DDP=1 wraps the model at N=1 too (hooks + bucket packing, no collective): a way to see
    # the data-parallel bookkeeping overhead on a one-GPU box.  Never the default.
    wrap = (lambda m: FlatDDP(m, single_rank_collectives=single_rank_rccl)) if (world > 1 or force_ddp) else None
    model, cfg, spec, step, (partial, gt, extra), B = make_workload(args.model, device, args.batch, args.views, rank, wrap)
    args.batch = B
    samples = B * args.views
    if args.graph == "auto":
        # The batch-16 image / segmentation steps are bound by the host's launch rate and are replayed from a HIP graph
        # (utils/graph_runner.py; falls back to eager by itself when its capture-time check fails).  The headline bs-32 step
        # is device-bound: measured per step with events on one box, replays 30.1 ms vs 30.05 eager, and the eager step that
        # has to carry the per-kernel events costs the graphed run 12 ms of transients (profiles/round4/README.md) - it stays
        # eager.  N > 1 stays eager as well: the bucket hooks launch RCCL from Python.
        args.graph = "on" if (wrap is None and world == 1 and args.model != "adapointr") else "off"
    eager_step = step
    if args.graph == "on":
        if wrap is not None:
            raise SystemExit("--graph on: the data-parallel wrapper launches collectives from Python hooks; one process only")
        step = GraphedTrainStep(step, warmup=3, verify=True)

    probe = ChamferProbe(N_GT, N_GT)
    _native.chamfer_forward = probe
    aprobe = AttentionProbe()
    _native.attention_forward, _native.attention_backward = aprobe.forward, aprobe.backward
    pprobe = PrimitiveProbe()
    _native.furthest_point_sample, _native.knn = pprobe.furthest_point_sample, pprobe.knn_call
    gprobe = GemmProbe()

    def barrier():
        if world > 1 or single_rank_rccl:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    note("model built; warm-up")
    graph_warmup = 0
    for i in range(args.warmup):
        eager_step(partial, gt, extra)
        if i == 0:
            torch.cuda.synchronize()
            note("first step done")
    if args.graph == "on":           # untimed: three eager steps in the graph's own workspaces, capture + check, one replay
        while graph_warmup < 8 and step.enabled and step.replays < 1:
            step(partial, gt, extra)
            graph_warmup += 1
        note(f"graphed step: enabled={step.enabled} after {graph_warmup} more untimed steps, capture check {step.verified}")
    barrier()
    note("timed region")
    probe.enabled = aprobe.enabled = pprobe.enabled = gprobe.enabled = True
    graphed = args.graph == "on" and step.enabled
    t0 = time.perf_counter()
    last = None
    marks = []                                  # long runs: one event per 100 steps, for the steady-state picture
    per_step = os.environ.get("PTOPS_BENCH_STEP_EVENTS") == "1"      # diagnostics: one event per step
    step_marks = []
    for i in range(args.steps):
        if per_step:
            step_marks.append(torch.cuda.Event(enable_timing=True))
            step_marks[-1].record()
        if args.steps >= 200 and i % 100 == 0:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
        # the first timed step is always an eager one: it carries the per-kernel events (a replay runs no Python)
        last = (eager_step if (i == 0 or not graphed) else step)(partial, gt, extra)
        aprobe.enabled = pprobe.enabled = gprobe.enabled = False   # ~270 timing events per step cost ~1 ms: only the first timed step carries them
    if marks:
        marks.append(torch.cuda.Event(enable_timing=True))
        marks[-1].record()
    if per_step:
        step_marks.append(torch.cuda.Event(enable_timing=True))
        step_marks[-1].record()
    barrier()
    if per_step:
        note("ms per timed step: " + " ".join(f"{a.elapsed_time(b):.2f}" for a, b in zip(step_marks[:-1], step_marks[1:])))
    dt = time.perf_counter() - t0
    note(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step")
    probe.enabled = aprobe.enabled = False
    _native.chamfer_forward = probe.orig
    _native.attention_forward, _native.attention_backward = aprobe.fwd, aprobe.bwd
    _native.furthest_point_sample, _native.knn = pprobe.fps, pprobe.knn
    gprobe.restore()

    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    losses = [float(x) * 1000 for x in last]

    if rank == 0:
        out = {
            "metric": (f"clouds/sec (train step, bs={B}/GPU), AdaPoinTr PCN 2048->16384" if args.model == "adapointr" else
                       f"clouds/sec (train step, bs={B}/GPU" + (f" x {args.views} views" if args.views > 1 else "") +
                       f"), {spec['name']} PCN 2048->16384"),
            "value": round(world * samples * args.steps / dt, 3), "unit": "clouds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: ranks share one GPU, gloo)" if rehearsal else ""),
            "config": {"workload": f"{spec['name']} {spec['yaml']}, PCN 2048->16384, bs={B}/GPU"
                                   + (", one 3x224x224 render per sample" if spec["img"] and args.views == 1 else "")
                                   + (f", {args.views} renders per object folded into the batch ([{B},{args.views},3,224,224] -> "
                                      f"{samples} samples per step; grouper + point encoder once per object; a cloud = one "
                                      "(object, view) sample)" if args.views > 1 else "")
                                   + (", 16-class one-hot + frozen GDANet segmentator" if spec["cls"] else "")
                                   + f", full train step (BASELINE {spec['config']})",
                       "global_batch": world * B, "samples_per_step": world * samples, "parallelism": f"dp{world}", "gemm_selection": gemm,
                       "gemm_arithmetic": ("fp32 in / fp32 out everywhere; Linear forward, input-gradient and the larger weight-gradient "
                                           "GEMMs with >= 16384 rows run on libptops' split-bf16 kernels (three bf16 pieces per operand, six exact piece "
                                           "products, fp32 accumulation: error vs float64 of the order of the fp32 MFMA library kernels' - asserted "
                                           "within 2x (forward / input gradient; floor 2e-6 of the output range) and 4x (weight gradients; floor 1e-5), "
                                           "tests/test_kernels_gpu.py::test_gemm_split3_is_fp32_accurate); PTOPS_SPLIT3_GEMM=0 "
                                           "keeps all GEMMs on hipBLASLt / rocBLAS fp32 MFMA (+0.8 ms per step)")
                       if TU.USE_SPLIT3_GEMM else "fp32 MFMA library kernels (hipBLASLt / rocBLAS) for every GEMM",
                       "params": sum(p.numel() for p in model.parameters())},
            "last_losses_x1000": {"denoise": round(losses[0], 4), "recon": round(losses[1], 4)},
            "roofline": probe.summary(),
            "roofline_attention": aprobe.summary(1),
            "roofline_gemm": gprobe.summary(),
            "roofline_primitives": pprobe.summary(),
        }
        if marks:
            n_last = args.steps - 100 * (len(marks) - 2)
            out["ms_per_step_by_100_steps"] = [round(a.elapsed_time(b) / (100 if i < len(marks) - 2 else n_last), 3)
                                               for i, (a, b) in enumerate(zip(marks[:-1], marks[1:]))]
        out["graphed"] = ({"enabled": step.enabled, "replays": step.replays, "capture_check": step.verified,
                           "disabled_reason": step.disabled_reason, "untimed_graph_warmup_steps": graph_warmup,
                           "note": "timed steps 2..K are replays of one HIP graph of the whole step "
                                   "(utils/graph_runner.py::GraphedTrainStep); timed step 1 runs eagerly and carries the "
                                   "per-kernel events of the roofline entries"}
                          if args.graph == "on" else {"enabled": False})
        if world == 1 and args.model == "adapointr" and args.views == 1 and not args.no_variants and not force_ddp:
            # the other single-GPU configurations of BASELINE.json, so that the driver's record carries them
            del step, eager_step, model
            torch.cuda.empty_cache()
            out["variants"] = {w: run_variant(w, device, note=note) for w in ("img", "seg", "adapointr")}
            out["variants"]["note"] = ("10 timed steps each after the headline region, one process: eager TrainStep, then the same "
                                       "step replayed from a HIP graph (utils/graph_runner.py::GraphedTrainStep, checked against "
                                       "the eager step at capture time); `adapointr` = the headline workload replayed, for "
                                       "comparison with the eager headline above (which carries the per-kernel probes)")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, spec)
        print(json.dumps(out), flush=True)
    if world > 1 or single_rank_rccl:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
