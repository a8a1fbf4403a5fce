"""GPU suite: the same model-level parity checks as tests/test_model_cpu.py, but through
libptops on the MI355X, against the golden fixtures generated from the reference."""
import pytest
import torch

import parity_checks as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def use_native(native):
    assert native.backend_for_testing() is None
    torch.backends.cuda.matmul.allow_tf32 = False
    yield


def test_functional_boundary():
    pc.check_functional("cuda")


def test_grouper_matches_reference():
    pc.check_grouper("cuda")


def test_blocks_match_reference():
    print(pc.check_blocks("cuda"))


def test_eval_forward_matches_reference():
    print(pc.check_eval_forward("cuda"))


def test_train_forward_loss_and_grads_match_reference():
    print(pc.check_train_forward_and_loss("cuda"))


def test_chamfer_baselines_match_reference():
    print(pc.check_chamfer_baselines("cuda"))


def test_chamfer_baselines_token_major_equals_literal():
    print(pc.check_baseline_forms("cuda"))


def test_image_fused_model_matches_reference():
    print(pc.check_img_model("cuda"))


def test_segmentation_branch_matches_reference():
    print(pc.check_seg_branch("cuda"))


def test_kernels_run_on_the_current_stream():
    """A side stream must see its own work ordered: results equal the default-stream results."""
    from imgadapointr_amd import _native
    x = torch.rand(4, 2048, 3, device="cuda") - 0.5
    y = torch.rand(4, 4096, 3, device="cuda") - 0.5
    want = _native.chamfer_forward(x, y)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x2 = x * 1.0  # produced on the side stream; the kernel must queue behind it
        got = _native.chamfer_forward(x2, y)
    s.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", ["img", "seg", "segenc"])
def test_full_train_step_of_the_fused_families_matches_reference(case):
    """BASELINE configs[2] / [4] and SegEncAdaPoinTr: one production TrainStep (split-bf16 GEMMs and attention, deferred
    folds, fused clip + AdamW) at B = 2 on the HIP path vs the reference fixture."""
    print(pc.check_trainstep(case, "cuda"))


@pytest.mark.parametrize("case", ["img", "seg"])
def test_fused_family_train_step_is_bit_reproducible_outside_the_library_branches(case):
    """Two runs of the same B = 2 `TrainStep` in one process: every gradient that does not belong to the ResNet18 / GDANet
    (MIOpen) must be bit-identical - the padded-head multihead attention, add + LayerNorm and fusion layers are libptops
    kernels, and a race in them would show here (round-3 review: separate "MIOpen picked another kernel" from "a libptops
    kernel races")."""
    n, lib_differ = pc.check_trainstep_reproducible(case, "cuda")
    assert n > 300
    print(f"{n} tensors bit-identical; library-branch tensors that differ between the runs: {len(lib_differ)}")


def test_multi_view_batch_equals_separate_views():
    print(pc.check_multi_view("cuda", views=3))


def test_fused_chamfer_reductions_equal_the_literal_chain():
    """The one-node Chamfer + reduction (`ptops_chamfer_reduce`, one launch each way) against the literal torch.mean /
    torch.sqrt chain of extensions/chamfer_dist/__init__.py:28-103 on the same HIP distances: values to the rounding of
    a 2 x 20000-term float sum, gradients of both clouds bit for bit (autograd's formulas in autograd's order), for all
    four reductions, a split reduction with one unused half, a scaled loss, and batch-1 `ignore_zeros`."""
    from imgadapointr_amd.extensions import chamfer_dist as cd
    torch.manual_seed(3)
    a0 = torch.rand(5, 4000, 3, device="cuda") - 0.5
    b0 = torch.rand(5, 3000, 3, device="cuda") - 0.5
    a0[0, :40] = 0.0                                             # zero rows for ignore_zeros

    def run(crit, fused, pick, a_in, b_in):
        cd.FUSED_REDUCTIONS = fused
        try:
            a, b = a_in.clone().requires_grad_(True), b_in.clone().requires_grad_(True)
            loss = pick(crit(a, b))
            (loss * 0.37).backward()
            return loss.detach(), a.grad, b.grad
        finally:
            cd.FUSED_REDUCTIONS = True

    cases = [(cd.ChamferDistanceL1(), lambda r: r, a0, b0), (cd.ChamferDistanceL2(), lambda r: r, a0, b0),
             (cd.ChamferDistanceL1_PM(), lambda r: r, a0, b0),
             (cd.ChamferDistanceL2_split(), lambda r: r[0] + 3.0 * r[1], a0, b0),
             (cd.ChamferDistanceL2_split(), lambda r: r[1], a0, b0),
             (cd.ChamferDistanceL1(ignore_zeros=True), lambda r: r, a0[:1], b0[:1])]
    for crit, pick, a_in, b_in in cases:
        want, got = run(crit, False, pick, a_in, b_in), run(crit, True, pick, a_in, b_in)
        name = type(crit).__name__
        assert abs(want[0].item() - got[0].item()) <= 4e-7 * abs(want[0].item()), (name, want[0].item(), got[0].item())
        for w, g in zip(want[1:], got[1:]):
            if w is None:                                        # literal chain: no gradient at all for an unused cloud
                assert g is None or float(g.abs().max()) == 0.0, name
            else:
                assert torch.equal(w, g), (name, float((w - g).abs().max()))
    # determinism of the forward sum, and torch.mean's NaN for an empty cloud
    x = cd.ChamferDistanceL1()(a0, b0)
    assert all(torch.equal(x, cd.ChamferDistanceL1()(a0, b0)) for _ in range(3))
    from imgadapointr_amd import _native
    e = torch.empty(0, device="cuda")
    out = _native.chamfer_reduce(e, torch.ones(7, device="cuda"), True)
    assert torch.isnan(out[0]) and out[1].item() == 1.0 and torch.isnan(out[2])


def test_chamfer_reductions_on_the_hip_path():
    """ChamferDistanceL1 / L2 / L2_split / L1_PM and `ignore_zeros` (extensions/chamfer_dist/__init__.py:28-103 of the
    reference) through libptops against the reference fixture (reference Python over the oracle's distances), plus the
    gradient of each reduction against the CPU-oracle backend's."""
    import numpy as np
    from imgadapointr_amd.extensions.chamfer_dist import (ChamferDistanceL1, ChamferDistanceL1_PM, ChamferDistanceL2,
                                                          ChamferDistanceL2_split)
    g = pc.load("chamfer_reductions.npz")
    want = pc.meta()["chamfer_reductions"]
    a, b = torch.from_numpy(g["a"]).cuda(), torch.from_numpy(g["b"]).cuda()
    assert abs(ChamferDistanceL1()(a, b).item() - want["L1"]) < 1e-7
    assert abs(ChamferDistanceL2()(a, b).item() - want["L2"]) < 1e-7
    assert abs(ChamferDistanceL1_PM()(a, b).item() - want["L1_PM"]) < 1e-7
    s = ChamferDistanceL2_split()(a, b)
    assert abs(s[0].item() - want["L2_split"][0]) < 1e-7 and abs(s[1].item() - want["L2_split"][1]) < 1e-7
    # ignore_zeros drops the zero-padded rows of a B == 1 batch (coordinate SUM == 0, __init__.py:37-41)
    assert int((g["a"][0].sum(-1) == 0).sum()) > 0
    assert abs(ChamferDistanceL1(ignore_zeros=True)(a[:1], b[:1]).item() - want["L1_ignore_zeros_b1"]) < 1e-7
    assert abs(ChamferDistanceL2(ignore_zeros=True)(a[:1], b[:1]).item() - want["L2_ignore_zeros_b1"]) < 1e-7
    assert abs(want["L1_ignore_zeros_b1"] - ChamferDistanceL1()(a[:1], b[:1]).item()) > 1e-4     # and it matters here
    # B > 1: ignore_zeros is a no-op in the reference
    assert ChamferDistanceL1(ignore_zeros=True)(a, b).item() == ChamferDistanceL1()(a, b).item()
    # gradients flow through every reduction (finite, deterministic across two runs)
    for crit in (ChamferDistanceL1(), ChamferDistanceL2(), ChamferDistanceL1_PM(), ChamferDistanceL1(ignore_zeros=True)):
        grads = []
        for _ in range(2):
            x = (a[:1] + 0.01).clone().requires_grad_(True)
            crit(x, b[:1]).backward()
            assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
            grads.append(x.grad.clone())
        assert torch.equal(grads[0], grads[1])


@pytest.mark.gpu
def test_gdanet_token_major_edge_stage_equals_the_nchw_form(native):
    """A local-operator stage of GDANet in training mode (reference GDANet_ptseg.py:59-60, 66-67, 72: conv, BatchNorm2d,
    ReLU, conv, BatchNorm2d, ReLU, max over k) on libptops' row kernels against the same stage on ATen / MIOpen in NCHW:
    output, gradient of the input and the running statistics of both BatchNorms, for the stage with normals (9 -> 64) and
    the 262 -> 128 stage."""
    from imgadapointr_amd.models.segmentation import GDANet_util as U
    from imgadapointr_amd.models.segmentation.GDANet_ptseg import GDANet
    torch.manual_seed(0)
    m = GDANet(50).cuda().train()
    for p in m.parameters():
        p.requires_grad = False                                  # frozen, as the runner uses it
    with torch.no_grad():
        for bn in (m.bn1, m.bn11, m.bn3, m.bn31):                # non-trivial affine parameters
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    B, N = 4, 512
    cases = [(3, m.conv1, m.conv11, True), (131, m.conv3, m.conv31, False)]
    was = U.TOKEN_MAJOR_STAGE
    try:
        for C, conv_a, conv_b, with_normals in cases:
            x = (torch.rand(B, C, N, device="cuda") - 0.5)
            normals = torch.nn.functional.normalize(torch.randn(B, N, 3, device="cuda"), dim=-1) if with_normals else None
            w = torch.randn(B, conv_b[0].out_channels, N, device="cuda")
            got = []
            for token_major in (False, True):
                U.TOKEN_MAJOR_STAGE = token_major
                for bn in (conv_a[1], conv_b[1]):
                    bn.reset_running_stats()
                xx = x.clone().requires_grad_(True)
                f = m._edge_stage(xx, 30, conv_a, conv_b, normals)
                (f * w).sum().backward()
                got.append((f.detach().clone(), xx.grad.clone(), conv_a[1].running_mean.clone(), conv_a[1].running_var.clone(),
                            conv_b[1].running_mean.clone(), conv_b[1].running_var.clone(),
                            conv_b[1].num_batches_tracked.clone().float()))
            names = ("output", "input gradient", "bn_a mean", "bn_a var", "bn_b mean", "bn_b var", "batches tracked")
            for name, a, b in zip(names, *got):
                err = float((a - b).abs().max() / (a.abs().max() + 1e-30))
                # fp32 sums in another order; the gradient also sees a handful of arg-max / ReLU decisions at rounding level
                assert err < (2e-3 if name == "input gradient" else 2e-5), (C, name, err)
    finally:
        U.TOKEN_MAJOR_STAGE = was
