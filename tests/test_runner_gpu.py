"""run_net on the GPU: the full-size AdaPoinTr of cfgs/PCN_models/AdaPoinTr.yaml on a synthetic PCN-shaped dataset,
one short epoch through the native kernels, validation (Chamfer / F-Score / EMD metrics), checkpoint, resume."""
import argparse
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_run_net_full_model_on_gpu(tmp_path, native):
    from imgadapointr_amd.tools import runner
    from imgadapointr_amd.utils.config import EasyDict, cfg_from_yaml_file
    from imgadapointr_amd.utils.metrics import Metrics
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        cfg = cfg_from_yaml_file("cfgs/PCN_models/AdaPoinTr.yaml")
    finally:
        os.chdir(cwd)
    ds = dict(NAME="SyntheticPCN", N_POINTS=16384, N_PARTIAL=2048, LENGTH=4)
    cfg.dataset = EasyDict(train=dict(_base_=dict(ds), others=dict(subset="train", bs=2)),
                           val=dict(_base_=dict(ds, LENGTH=2), others=dict(subset="val")),
                           test=dict(_base_=dict(ds, LENGTH=2), others=dict(subset="test")))
    cfg.total_bs = 2
    cfg.max_epoch = 0          # epoch 0 only
    args = argparse.Namespace(distributed=False, world_size=1, local_rank=0, use_gpu=True, num_workers=0, resume=False,
                              start_ckpts=None, ckpts=None, val_freq=1, sync_bn=False, log_every=1,
                              experiment_path=str(tmp_path), log_name="gpu_run", test=False)
    torch.manual_seed(0)
    best = runner.run_net(args, cfg)
    assert isinstance(best, Metrics)
    vals = best.state_dict()
    assert set(vals) == {"F-Score", "CDL1", "CDL2", "EMDistance"} and all(v == v for v in vals.values())   # no NaN
    for name in ("ckpt-best.pth", "ckpt-last.pth"):
        assert os.path.exists(tmp_path / name), name
    ck = torch.load(tmp_path / "ckpt-last.pth", weights_only=False)
    assert ck["epoch"] == 0 and len(ck["base_model"]) > 100
    # resume picks the checkpoint up and runs one more epoch on the GPU
    cfg.max_epoch = 1
    args.resume = True
    best2 = runner.run_net(args, cfg)
    assert isinstance(best2, Metrics)
    assert torch.load(tmp_path / "ckpt-last.pth", weights_only=False)["epoch"] == 1
    # --graph_train: one more epoch of six iterations (three eager, one captured + checked, two replays) through HIP graphs
    from imgadapointr_amd.utils import graph_runner
    made = []
    orig = graph_runner.GraphedTrainStep.__init__
    graph_runner.GraphedTrainStep.__init__ = lambda self, *a, **k: (orig(self, *a, **k), made.append(self))[0]
    try:
        cfg.dataset.train._base_.LENGTH = 12
        cfg.max_epoch = 2
        args.graph_train = True
        best3 = runner.run_net(args, cfg)
    finally:
        graph_runner.GraphedTrainStep.__init__ = orig
    assert isinstance(best3, Metrics) and all(v == v for v in best3.state_dict().values())
    assert len(made) == 1 and made[0].enabled and made[0].replays >= 2, (made[0].disabled_reason, made[0].replays)
    ck = torch.load(tmp_path / "ckpt-last.pth", weights_only=False)
    assert ck["epoch"] == 2 and {float(s["step"]) for s in ck["optimizer"]["state"].values()} == {2.0 + 2.0 + 6.0}


@pytest.mark.gpu
def test_graphed_forward_matches_eager(native):
    """Validation forward replayed as one HIP graph: bit-identical to eager, refreshed inputs, eager fallback."""
    import parity_checks as pc
    from imgadapointr_amd.utils.graph_runner import GraphedForward
    model, _ = pc.build_adapointr("cuda")
    model.eval()
    gf = GraphedForward(model)
    torch.manual_seed(3)
    with torch.no_grad():
        for i in range(3):
            x = (torch.rand(1, 2048, 3, device="cuda") - 0.5)
            want = model(x)
            got = gf(x)
            assert all(torch.equal(a, b) for a, b in zip(got, want)), i
            # what validation does between two clouds: other kernels, allocations and a host synchronisation
            # (metrics .item()).  A graph with a MEMSET node returns stale outputs after this (csrc/core.hip).
            junk = torch.zeros(1000, device="cuda") + float(want[1].sum().item())
            torch.cuda.synchronize()
        assert gf.enabled and gf.disabled_reason is None      # the eval forward is graph-safe: no memset node in it
        assert gf.replays == 3
        xb = torch.rand(2, 2048, 3, device="cuda") - 0.5         # a second input signature gets its own graph
        assert all(torch.equal(a, b) for a, b in zip(gf(xb), model(xb)))
        assert gf.replays == 4
    x = torch.rand(1, 2048, 3, device="cuda") - 0.5
    out = gf(x)                                                   # gradients enabled: plain eager call
    assert gf.replays == 4 and out[-1].requires_grad


@pytest.mark.gpu
def test_graphed_forward_guards(native):
    """LRU cap on captured signatures, eager fallback with a stated reason when the forward cannot be captured,
    and the replay-vs-eager check that rejects a graph which does not reproduce the forward."""
    from imgadapointr_amd.utils.graph_runner import GraphedForward

    class Scale(torch.nn.Module):
        def forward(self, x):
            return x * 2.0 + 1.0

    gf = GraphedForward(Scale().cuda().eval(), max_graphs=1)
    with torch.no_grad():
        for n in (8, 16, 8, 16):                                  # two signatures, one slot: re-captured, never stale
            x = torch.rand(n, 3, device="cuda")
            assert torch.equal(gf(x), x * 2.0 + 1.0)
            assert len(gf._cache) == 1
    assert gf.enabled and gf.replays == 4

    class HostSync(torch.nn.Module):
        def forward(self, x):
            return x * float(x.sum().item())                      # device-to-host read: not capturable

    gs = GraphedForward(HostSync().cuda().eval())
    with torch.no_grad():
        x = torch.rand(4, 3, device="cuda")
        assert torch.allclose(gs(x), x * x.sum())
        assert not gs.enabled and "capture failed" in gs.disabled_reason
        assert torch.allclose(gs(x), x * x.sum())                 # stays eager afterwards

    class Stateful(torch.nn.Module):                              # every call returns something else: a replay cannot
        def __init__(self):                                       # reproduce the eager warm-up -> graphing is refused
            super().__init__()
            self.register_buffer("calls", torch.zeros(()))

        def forward(self, x):
            self.calls += 1
            return x + self.calls

    class StridedSum(torch.nn.Module):                            # ATen zeroes this reduction's semaphores with a memset
        def forward(self, x):
            return x.sum(dim=1)

    gm = GraphedForward(StridedSum().cuda().eval())
    with torch.no_grad():
        for _ in range(3):
            x = torch.randn(1, 512, 1024, device="cuda")
            assert torch.equal(gm(x), x.sum(dim=1))               # right either way: graphed and sound, or refused
            torch.cuda.synchronize()
    assert gm.enabled or "host synchronisation" in gm.disabled_reason

    gd = GraphedForward(Stateful().cuda().eval())
    with torch.no_grad():
        gd(torch.zeros(4, device="cuda"))
    assert not gd.enabled and "did not reproduce" in gd.disabled_reason


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_flat_ddp_call_sequence_on_rccl(native):
    """The multi-GPU path's RCCL calls, run with one rank on this box (tests/rccl_single_rank.py); a child process
    with its own time limit, so a wedged rendezvous fails instead of hanging the suite."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_PORT="29637")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    if r.returncode != 0 or "rccl single-rank ok" not in r.stdout:
        print(r.stdout[-2000:])
        print(r.stderr[-6000:])
    assert r.returncode == 0 and "rccl single-rank ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_fast_paths_single_step_all_gradients(native):
    """ONE production train step of AdaPoinTr at the bench configuration (bs = 32, where every fast path is active:
    split-bf16 GEMMs incl. weight gradients and Linear + GELU, split-bf16 attention forward / backward, deferred folds,
    fused clip + AdamW) against the same step with all of them switched off (library GEMMs, f32-MFMA attention, immediate
    folds, clip_grad_norm_ + AdamW.step): both losses, EVERY parameter gradient tensor by tensor, and every parameter
    after the update.  A 30-step loss trajectory cannot tell a small systematic error in one kernel from chaotic drift
    (round-2 review); the first step can."""
    import parity_checks as pc
    from imgadapointr_amd.models import Transformer_utils as TU
    from imgadapointr_amd.tools import builder, runner
    from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm
    enable_tuned_gemm()
    B = 32
    torch.manual_seed(0)
    partial = (torch.rand(B, 2048, 3) - 0.5).cuda()
    gt = (torch.rand(B, 16384, 3) - 0.5).cuda()
    noise = (torch.randn(B, 64, 3) * 0.01).clamp_(-0.05, 0.05).cuda()
    saved = (TU.USE_SPLIT3_GEMM, TU.USE_SPLIT3_WGRAD, TU.FUSED_LINEAR_GELU, TU.USE_MFMA_ATTENTION_BWD, runner.DEFER_FOLDS,
             runner.FUSED_CLIP_ADAMW)

    def run(fast):
        (TU.USE_SPLIT3_GEMM, TU.USE_SPLIT3_WGRAD, TU.FUSED_LINEAR_GELU, TU.USE_MFMA_ATTENTION_BWD, runner.DEFER_FOLDS,
         runner.FUSED_CLIP_ADAMW) = saved if fast else (False, False, False, "fused", False, False)
        native.set_option("PTOPS_ATTN_FWD_F32", 0 if fast else 1)
        model, cfg = pc.build_adapointr("cuda")
        model.train()
        named = dict(model.named_parameters())
        step = runner.TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
        grads = {}
        step.before_update = lambda: grads.update({n: p.grad.detach().clone() for n, p in named.items() if p.grad is not None})
        before = {n: p.detach().clone() for n, p in named.items()}
        l0, l1 = step(partial, gt, (), 0, denoise_noise=noise)
        return (l0.item(), l1.item()), grads, before, {n: p.detach().clone() for n, p in named.items()}

    try:
        (fa, ga, _, pa), (fb, gb, before, pb) = run(True), run(False)
    finally:
        (TU.USE_SPLIT3_GEMM, TU.USE_SPLIT3_WGRAD, TU.FUSED_LINEAR_GELU, TU.USE_MFMA_ATTENTION_BWD, runner.DEFER_FOLDS,
         runner.FUSED_CLIP_ADAMW) = saved
        native.set_option("PTOPS_ATTN_FWD_F32", 0)
    # both forms are fp32-accurate; what moves the loss between them is a handful of arg-max decisions (query ranking, max-pools)
    # taken on rounding-level differences: 2.0e-5 relative was read on two of ten boxes (the former bound, missed by 1e-9),
    # typical 5e-6; a wrong kernel shows at 1e-3 and above
    assert abs(fa[0] - fb[0]) < 1e-4 * abs(fb[0]) and abs(fa[1] - fb[1]) < 1e-4 * abs(fb[1]), (fa, fb)
    assert set(ga) == set(gb) and len(ga) == 332 - 8          # 335 state-dict tensors - 3 BatchNorm buffers - 8 unused
    total = sum(float(v.double().pow(2).sum()) for v in gb.values()) ** 0.5
    rel, sq_err = {}, 0.0
    for n in gb:
        ref_norm = float(gb[n].double().norm())
        err = float((ga[n].double() - gb[n].double()).norm())
        sq_err += err * err
        if ref_norm < 1e-5 * total:                     # gradients that are zero in exact arithmetic (k_map biases ...)
            assert float(ga[n].double().norm()) < 1e-4 * total, n
            continue
        rel[n] = err / ref_norm
    worst = max(rel.items(), key=lambda kv: kv[1])
    over = sorted(((v, n) for n, v in rel.items() if v >= 2e-2), reverse=True)
    # rounding-level differences flip a few arg-maxes (max-pools, kNN-free top-k of the query ranking), which moves single
    # tensors by ~1 % (tests/parity_checks.py) and once in eight boxes of the pool one of them past 2 %; a wrong kernel moves
    # ITS tensors by O(1) and the whole gradient with them: at most three tensors may pass 2 %, none 15 %, and the error of
    # the whole gradient vector stays below 2.5 % (0.7 % on a typical box, worst tensor 1.6 %)
    assert len(rel) >= 280 and len(over) <= 3 and worst[1] < 0.15 and sq_err ** 0.5 < 2.5e-2 * total, \
        (worst, over[:6], sq_err ** 0.5 / total, len(rel))
    lr = 1e-4
    far = n_el = 0
    for n in pb:
        if n not in gb:
            assert torch.equal(pa[n], before[n]) and torch.equal(pb[n], before[n]), n       # no gradient: untouched
            continue
        d = (pa[n] - pb[n]).abs()
        far += int((d > 0.05 * lr).sum())
        n_el += d.numel()
        assert float(d.max()) <= 2.2 * lr, (n, float(d.max()))
    assert far <= 0.03 * n_el, (far, n_el)              # first AdamW step = lr * sign(g): only near-zero gradients differ
    print("worst gradient tensor", worst, "whole-gradient error", sq_err ** 0.5 / total, "elements off after the update", far / n_el)


@pytest.mark.gpu
def test_train_step_gradients_are_bit_reproducible(native):
    """Two runs of the same AdaPoinTr train step (same weights, inputs and jitter) give BIT-IDENTICAL gradients for every
    parameter: the scatter-add backward passes of the path sum in a fixed order (Chamfer over the transposed neighbour
    map, edge-conv over the transposed kNN graph) or in fixed point (gather-max), where the reference's kernels use float
    atomics in arrival order (extensions/chamfer_dist/chamfer.cu:193-198)."""
    import parity_checks as pc
    from imgadapointr_amd.tools import builder, runner
    torch.manual_seed(0)
    B = 4
    partial = (torch.rand(B, 2048, 3) - 0.5).cuda()
    gt = (torch.rand(B, 16384, 3) - 0.5).cuda()
    noise = (torch.randn(B, 64, 3) * 0.01).clamp_(-0.05, 0.05).cuda()

    def run():
        model, cfg = pc.build_adapointr("cuda")
        model.train()
        named = dict(model.named_parameters())
        step = runner.TrainStep(model, builder.build_optimizer(model, cfg), 10, 1)
        grads = {}
        step.before_update = lambda: grads.update({n: p.grad.detach().clone() for n, p in named.items() if p.grad is not None})
        losses = step(partial, gt, (), 0, denoise_noise=noise)
        return [float(v) for v in losses], grads, {n: p.detach().clone() for n, p in named.items()}

    (la, ga, pa), (lb, gb, pb) = run(), run()
    assert la == lb
    diff = [n for n in ga if not torch.equal(ga[n], gb[n])]
    assert not diff, diff[:8]
    assert all(torch.equal(pa[n], pb[n]) for n in pa)


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("case", ["adapointr", "img", "seg", "segenc"])
def test_graphed_train_step_equals_eager(native, case):
    """`GraphedTrainStep` (the whole train step replayed from one HIP graph; utils/graph_runner.py) against the eager
    `TrainStep` from identical weights over seven steps with fresh inputs each step, an LR change and host
    synchronisations in between (the sequence that exposed unreliable graph nodes on this stack in round 2): the first
    three steps run eagerly inside the graph's own workspaces, the fourth is captured and checked against the eager step
    by the object itself, the rest are replays.  Losses and all parameters must agree bit for bit where every kernel is
    this build's (AdaPoinTr), and to the run-to-run spread of the library kernels for the image / segmentation branches."""
    import parity_checks as pc
    from imgadapointr_amd.tools import builder, runner
    from imgadapointr_amd.utils.graph_runner import GraphedTrainStep
    B, steps = (4, 7) if case == "adapointr" else (2, 7)

    def build():
        if case == "adapointr":
            model, cfg = pc.build_adapointr("cuda")
            spec = dict(takes_img=False, has_img=False, has_cls=False, epoch=0)
        else:
            model, cfg, spec = pc.build_trainstep_model(case, "cuda")
        model.train()
        return model, runner.TrainStep(model, builder.build_optimizer(model, cfg), 10, 1), spec

    def batch(i, spec):
        g = torch.Generator().manual_seed(100 + i)
        partial = (torch.rand(B, 2048, 3, generator=g) - 0.5).cuda()
        gt = (torch.rand(B, 16384, 3, generator=g) - 0.5).cuda()
        extra = []
        if spec.get("has_img"):
            extra.append(torch.rand(B, 3, 224, 224, generator=g).cuda())
        elif spec.get("takes_img"):
            extra.append(None)
        if spec.get("has_cls"):
            cls = torch.zeros(B, 16)
            cls[torch.arange(B), torch.arange(B) % 16] = 1.0
            extra.append(cls.cuda())
        return partial, gt, tuple(extra)

    def run(graphed):
        model, step, spec = build()
        if graphed:
            step = GraphedTrainStep(step, warmup=3, verify=True)
        losses = []
        for i in range(steps):
            if i == 5:
                for grp in step.optimizer.param_groups:
                    grp["lr"] *= 0.5
            torch.manual_seed(1000 + i)                      # the jitter (and GDANet's dropout) draw from here in both runs
            partial, gt, extra = batch(i, spec)
            l0, l1 = step(partial, gt, extra, spec["epoch"])
            losses.append((l0.item(), l1.item()))            # a host synchronisation between steps
        step.optimizer.state_dict()                          # step counters back into torch's format
        return step, losses, {n: p.detach().clone() for n, p in model.named_parameters()}, \
            [float(step.optimizer.state[p]["step"]) for p in model.parameters() if p in step.optimizer.state]

    with pc._Pinned():
        eager_step, le, pe, se = run(False)
        graph_step, lg, pg, sg = run(True)
    assert graph_step.enabled, graph_step.disabled_reason
    assert graph_step.replays == steps - 3, graph_step.replays          # the captured step's own replay + the following ones
    assert se == sg and set(se) == {float(steps)}
    print(case, "capture-time check:", graph_step.verified, "losses", le[-1], lg[-1])
    if case == "adapointr":
        assert set(graph_step.verified["replay_vs_eager"].values()) == {0.0}, graph_step.verified
        assert le == lg, (le, lg)
        assert all(torch.equal(pe[n], pg[n]) for n in pe)
    else:
        # seg: GDANet in train mode over B = 2 (feature-space top-k at near-ties, a max-pool backward without a deterministic
        # form) amplifies rounding-level differences over the steps (its arms differ from each other by
        # percents in the gradient), and its branch runs beside the decoder inside the graph; 3e-4 (literal local operator) and
        # 1.4e-2 (first convolution before the gather) were seen at steps 5-6
        tol = 1e-3 if case == "seg" else 1e-4
        if case == "segenc":
            # segmentation BEFORE the encoder output: GDANet's dropout draws from the generator before the denoise jitter
            # does, and the graphed step draws the jitter first (outside the graph) - both are valid steps on different random
            # numbers, so only the capture-time check (same numbers handed to both) and the size of the losses are compared
            tol = 0.25
            assert all(v <= 1e-2 for v in graph_step.verified["replay_vs_eager"].values()), graph_step.verified
        rel = [max(abs(a0 - b0) / abs(a0), abs(a1 - b1) / abs(a1)) for (a0, a1), (b0, b1) in zip(le, lg)]
        print(case, "per-step relative loss difference:", ["%.1e" % r for r in rel])
        for i, r in enumerate(rel):
            # seg: steps 0-3 are eager steps in both runs (the fourth is the capture's checked eager step) and must agree like
            # two eager runs do - bit for bit under the pinned library choices; this is what caught the memset probe of
            # utils/graph_runner.py drawing random numbers between the jitter and GDANet's dropout masks.  Step 4 is the first
            # replay, from an identical state.  Afterwards the two trajectories are two samples of a chaotic map (a replayed
            # update differs from the eager one at the 1e-6 level and GDANet's neighbour sets flip on that: 1e-2 by step 5-6)
            limit = tol if case != "seg" else (1e-6 if i <= 3 else 1e-4 if i == 4 else 5e-2)
            assert r <= limit, (i, rel, le, lg)
        lr = 1e-4
        far = sum(int(((pe[n] - pg[n]).abs() > 0.5 * lr).sum()) for n in pe)
        assert case == "segenc" or far <= 0.02 * sum(p.numel() for p in pe.values()), far


@pytest.mark.gpu
def test_validate_and_metrics_match_the_reference_on_the_hip_path(native):
    """validate() / Metrics.get on the GPU (Chamfer kernels, fused reductions, device-side F-Score) against the
    reference's own validate() on the same predictions (tests/golden/metrics_validate.json)."""
    import parity_checks as pc
    print(pc.check_validate("cuda"))


@pytest.mark.gpu
def test_inference_cli_matches_the_reference_on_the_hip_path(native, tmp_path):
    """tools/inference.py end to end on the GPU (graphed eval forward) against the reference's fine.npy."""
    import parity_checks as pc
    print(pc.check_inference("cuda", tmp_path))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_graphed_train_step_recaptures_when_a_baked_scalar_changes(native):
    """The epoch-dependent coarse-loss weight of the VariableLoss models and the BatchNorm momenta (changed by the runner's
    scheduler between epochs) are Python numbers that a captured launch sequence bakes in: `GraphedTrainStep` keys its
    graphs on them.  Image model, B = 2: five steps at epoch 0, six at epoch 1 (another loss weight), then three with another
    BatchNorm momentum - three graphs, each preceded by its own eager steps, losses equal to the eager run throughout."""
    import parity_checks as pc
    from imgadapointr_amd.tools import builder, runner
    from imgadapointr_amd.utils.graph_runner import GraphedTrainStep
    B = 2
    plan = [0] * 5 + [1] * 6 + [1] * 5          # epochs; the momentum changes before the last five steps

    def run(graphed):
        model, cfg, spec = pc.build_trainstep_model("img", "cuda")
        model.train()
        step = runner.TrainStep(model, builder.build_optimizer(model, cfg), 10, 1)
        assert model.coarse_loss_weight(0) != model.coarse_loss_weight(1)
        if graphed:
            step = GraphedTrainStep(step, warmup=3, verify=True, max_graphs=2)
        losses = []
        for i, epoch in enumerate(plan):
            if i == 11:
                for m in model.modules():
                    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                        m.momentum = 0.05
            torch.manual_seed(2000 + i)
            g = torch.Generator().manual_seed(300 + i)
            partial = (torch.rand(B, 2048, 3, generator=g) - 0.5).cuda()
            gt = (torch.rand(B, 16384, 3, generator=g) - 0.5).cuda()
            img = torch.rand(B, 3, 224, 224, generator=g).cuda()
            l0, l1 = step(partial, gt, (img,), epoch)
            losses.append((l0.item(), l1.item()))
        return step, losses, {n: b.detach().clone() for n, b in model.named_buffers() if "running_mean" in n}

    with pc._Pinned():
        _, le, be = run(False)
        gs, lg, bg = run(True)
    assert gs.enabled, gs.disabled_reason
    # per key: three eager steps, the capture (its check replays once), then replays
    assert gs.replays == (5 - 3) + (6 - 3) + (5 - 3), gs.replays
    assert len(gs._cache) == 2                                   # max_graphs = 2: the epoch-0 graph was dropped
    for (a0, a1), (b0, b1) in zip(le, lg):
        assert abs(a0 - b0) <= 1e-4 * abs(a0) and abs(a1 - b1) <= 1e-4 * abs(a1), (le, lg)
    # the changed momentum reached the replayed BatchNorm launches: running statistics follow the eager run
    for n in be:
        assert torch.allclose(be[n], bg[n], rtol=1e-3, atol=1e-5), n


@pytest.mark.gpu
def test_capture_does_not_tune_gemms(native):
    """A GEMM shape that TunableOp meets for the first time INSIDE a capture (in a captured train step a weight gradient on a
    side stream takes the library path, which the eager warm-up steps never did: `bench.py --model seg --views 6` lost its
    graph that way) must not be tuned there: tuning times candidates with events, which invalidates the capture."""
    from imgadapointr_amd.utils.gemm_tuning import enable_tuned_gemm
    from imgadapointr_amd.utils.graph_runner import GraphedForward
    enable_tuned_gemm()
    tun = torch.cuda.tunable
    if not (tun.is_enabled() and tun.tuning_is_enabled()):
        pytest.skip("TunableOp is not tuning in this configuration")

    class ShapeChangesUnderCapture(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(328, 200))

        def forward(self, x):
            w = self.w[:, :136] if torch.cuda.is_current_stream_capturing() else self.w      # 136: a K no layer of the path has
            return x[:, :w.size(1)] @ w.t()

    m = ShapeChangesUnderCapture().cuda().eval()
    gf = GraphedForward(m, verify=False)
    with torch.no_grad():
        x = torch.randn(1000, 200, device="cuda")
        for _ in range(3):
            out = gf(x)
        assert gf.enabled, gf.disabled_reason
        assert torch.allclose(out, x[:, :136] @ m.w[:, :136].t(), rtol=1e-4, atol=1e-4)
    assert tun.tuning_is_enabled()                            # restored after the capture
