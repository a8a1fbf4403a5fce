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


def test_multi_view_batch_equals_separate_views():
    print(pc.check_multi_view("cuda", views=3))


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
