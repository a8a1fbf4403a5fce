"""Model-level parity checks against the golden fixtures (generated from the reference's own
Python by tests/golden/make_golden.py).  Device-agnostic: the CPU tests run them on the oracle
backend, the GPU tests on libptops."""
import json
import os

import numpy as np
import torch

from golden_util import seeded_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def meta():
    with open(os.path.join(GOLD, "meta.json")) as fh:
        return json.load(fh)


def t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def build_adapointr(device):
    from imgadapointr_amd.models import build_model_from_cfg
    from imgadapointr_amd.utils.config import cfg_from_yaml_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cwd = os.getcwd()
    os.chdir(root)  # `_base_` paths in the YAML are relative to the repo root, as in the reference
    try:
        cfg = cfg_from_yaml_file("cfgs/PCN_models/AdaPoinTr.yaml")
    finally:
        os.chdir(cwd)
    model = build_model_from_cfg(cfg.model)
    model.load_state_dict(seeded_state_dict(model.state_dict(), "adapointr"))
    return model.to(device), cfg


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def check_functional(device):
    from imgadapointr_amd.models import Transformer_utils as TU
    g = load("functional.npz")
    xyz, new = t(g["xyz"], device), t(g["new_xyz"], device)
    # square_distance is pure torch in both; BLAS accumulation order may differ in the last ulp
    assert maxdiff(TU.square_distance(new, xyz).cpu().numpy(), g["square_distance"]) < 2e-6
    # kNN: the reference returns an unordered top-k whose membership on near-ties depends on
    # rounding; require set equality except where the k-th / (k+1)-th distances are within 1e-6
    got = np.sort(TU.knn_point(8, xyz, new).cpu().numpy(), -1)
    want = g["knn_sorted"]
    sq = g["square_distance"]
    for b in range(got.shape[0]):
        for s in range(got.shape[1]):
            if not np.array_equal(got[b, s], want[b, s]):
                diff = set(got[b, s]) ^ set(want[b, s])
                ds = [sq[b, s, j] for j in diff]
                assert max(ds) - min(ds) < 1e-6, (b, s, diff, ds)
    idx = t(g["knn_sorted"].astype(np.int64), device)
    assert np.array_equal(TU.index_points(t(g["points"], device), idx).cpu().numpy(),
                          _np_index(g["points"], g["knn_sorted"]))


def _np_index(points, idx):
    B = points.shape[0]
    return np.stack([points[b][idx[b]] for b in range(B)])


def check_grouper(device, tol=2e-5):
    from imgadapointr_amd.models.base_blocks import DGCNN_Grouper
    g = load("grouper.npz")
    m = DGCNN_Grouper(k=16)
    m.load_state_dict(seeded_state_dict(m.state_dict(), "grouper"))
    m = m.to(device).eval()
    with torch.no_grad():
        coor, f = m(t(g["x"], device), [128, 64])
    assert np.array_equal(coor.cpu().numpy(), g["coor"])  # FPS picks: same points, bit for bit
    assert maxdiff(f.cpu().numpy(), g["f"]) < tol * max(1.0, float(np.abs(g["f"]).max()))


def check_blocks(device, tol=3e-5):
    from imgadapointr_amd.models import Transformer_utils as TU
    from imgadapointr_amd.models import base_blocks as BB
    g = load("blocks.npz")
    dim, heads, k = 48, 6, 8
    q_pos, v_pos, q, v = (t(g[n], device) for n in ("q_pos", "v_pos", "q", "v"))
    S = q.shape[1]

    def mk(mod, tag):
        mod.load_state_dict(seeded_state_dict(mod.state_dict(), tag, std=0.2))
        return mod.to(device).eval()

    res = {}
    with torch.no_grad():
        for sdpa in (True, False):
            TU.USE_SDPA = sdpa
            attn = mk(TU.Attention(dim, num_heads=heads, qkv_bias=True), "attn")
            mask = torch.zeros(S, S, device=device)
            mask[:-16, -16:] = 1.
            res[f"attn_nomask/{sdpa}"] = (attn(q), g["attn_nomask"])
            res[f"attn_mask/{sdpa}"] = (attn(q, mask=mask), g["attn_mask"])
            ca = mk(TU.CrossAttention(dim, dim, num_heads=heads, qkv_bias=True), "cross")
            res[f"cross/{sdpa}"] = (ca(q, v), g["cross"])
        TU.USE_SDPA = True
        for fused in (True, False):
            TU.FUSED_GRAPH_ATTENTION = fused
            dga = mk(TU.DynamicGraphAttention(dim, k=k), "dga")
            res[f"dga_self/{fused}"] = (dga(q, q_pos), g["dga_self"])
            res[f"dga_cross/{fused}"] = (dga(q=q, v=v, q_pos=q_pos, v_pos=v_pos), g["dga_cross"])
            res[f"dga_denoise/{fused}"] = (dga(q, q_pos, denoise_length=16), g["dga_denoise"])
        TU.FUSED_GRAPH_ATTENTION = True
        enc = mk(BB.SelfAttnBlockApi(dim, heads, mlp_ratio=2., qkv_bias=True, block_style='attn-graph',
                                     combine_style='concat', k=k), "encblock")
        res["enc_block"] = (enc(q, q_pos, idx=TU.knn_idx(k, q_pos, q_pos)), g["enc_block"])
        dec = mk(BB.CrossAttnBlockApi(dim, heads, mlp_ratio=2., qkv_bias=True,
                                      self_attn_block_style='attn-graph', self_attn_combine_style='concat',
                                      cross_attn_block_style='attn-graph', cross_attn_combine_style='concat', k=k),
                 "decblock")
        res["dec_block_eval"] = (dec(q, v, q_pos, v_pos, self_attn_idx=TU.knn_idx(k, q_pos, q_pos),
                                     cross_attn_idx=TU.knn_idx(k, v_pos, q_pos)), g["dec_block_eval"])
        res["dec_block_denoise"] = (dec(q, v, q_pos, v_pos, self_attn_idx=None,
                                        cross_attn_idx=TU.knn_idx(k, v_pos, q_pos), denoise_length=16),
                                    g["dec_block_denoise"])
    worst = {}
    for name, (got, want) in res.items():
        d = maxdiff(got.cpu().numpy(), want)
        worst[name] = d
        assert d < tol * max(1.0, float(np.abs(want).max())), (name, d)
    return worst


def _match_rows(got, want):
    """Permutation p with got[i] ~ want[p[i]] (nearest row; must be a bijection)."""
    d = ((got[:, None, :].astype(np.float64) - want[None, :, :].astype(np.float64)) ** 2).sum(-1)
    p = d.argmin(1)
    assert len(set(p.tolist())) == len(p), "rows do not match one-to-one"
    return p


def compare_ranked(coarse, fine_head, want_coarse, want_fine_head, tol, factor=32):
    """max deviations of ranked-query outputs [B,M,3] / [B,K*factor,3] after matching query rows.
    `*_fine_head` hold the points of the first K queries; groups whose query kept its rank are compared."""
    dc = df = 0.0
    swapped = 0
    for b in range(coarse.shape[0]):
        p = _match_rows(coarse[b], want_coarse[b])
        swapped += int((p != np.arange(len(p))).sum())
        dc = max(dc, maxdiff(coarse[b], want_coarse[b][p]))
        K = fine_head.shape[1] // factor
        keep = np.nonzero(p[:K] == np.arange(K))[0]
        if len(keep):
            got = fine_head[b].reshape(K, factor, 3)[keep]
            want = want_fine_head[b].reshape(K, factor, 3)[keep]
            df = max(df, maxdiff(got, want))
    assert dc < tol and df < tol, (dc, df)
    assert swapped <= 8, swapped
    return dc, df, swapped


def check_eval_forward(device, tol=2e-4):
    """Eval forward vs the reference fixture.  The 512 queries are the top-ranked of 768 proposals
    (argsort of a learned score, AdaPoinTr.py:119-121); two scores closer than float rounding may swap
    places between implementations, so rows are matched before comparing (values must still agree)."""
    g = load("adapointr.npz")
    model, _ = build_adapointr(device)
    model.eval()
    with torch.no_grad():
        coarse, fine = model(t(g["partial"], device))
    coarse, fine = coarse.cpu().numpy(), fine.cpu().numpy()
    dc = df = 0.0
    swapped = 0
    for b in range(coarse.shape[0]):
        p = _match_rows(coarse[b], g["eval_coarse"][b])
        swapped += int((p != np.arange(len(p))).sum())
        dc = max(dc, maxdiff(coarse[b], g["eval_coarse"][b][p]))
        want_fine = g["eval_fine"][b].reshape(len(p), -1, 3)[p].reshape(-1, 3)
        df = max(df, maxdiff(fine[b], want_fine))
    assert dc < tol and df < tol, (dc, df)
    assert swapped <= 8, swapped
    return dc, df, swapped


def check_train_forward_and_loss(device, tol=2e-4):
    g = load("adapointr.npz")
    mt = meta()
    model, _ = build_adapointr(device)
    model.train()
    ret = model(t(g["partial"], device), denoise_noise=t(g["train_noise"], device))
    for r, n in zip(ret[1:3], ("denoised_coarse", "denoised_fine")):
        assert maxdiff(r.detach().cpu().numpy(), g[n]) < tol, n
    compare_ranked(ret[0].detach().cpu().numpy(), ret[3][:, :2048].detach().cpu().numpy(), g["pred_coarse"],
                   g["pred_fine_head"], tol)
    l_den, l_rec = model.get_loss(ret, t(g["gt"], device))
    # CD-L1 x1000 within 1e-3 of the reference path (BASELINE.json north_star)
    assert abs(l_den.item() * 1000 - float(g["loss_denoised"]) * 1000) < 1e-3
    assert abs(l_rec.item() * 1000 - float(g["loss_recon"]) * 1000) < 1e-3
    (l_den + l_rec).backward()
    no_grad = sorted(n for n, p in model.named_parameters() if p.grad is None)
    assert no_grad == mt["no_grad_params"]
    gsq = {n: float((p.grad.double() ** 2).sum()) for n, p in model.named_parameters() if p.grad is not None}
    total = sum(gsq.values())
    if os.environ.get("PTOPS_DUMP_GSQ"):
        import json
        with open(os.environ["PTOPS_DUMP_GSQ"], "w") as fh:
            json.dump(gsq, fh)
    # The model is only piecewise smooth (max-pools over neighbours / tokens, top-k query selection): rounding-level
    # differences flip a few arg-maxes, which moves an individual parameter's gradient norm by up to ~1 % (first
    # grouper layer).  Which way the near-ties of this fixture fall depends on the GEMM kernels the BLAS library picks:
    # two MI355X boxes gave totals of 3.30603 (-0.05 %) and 3.29955 (-0.24 %), each bit-reproducible on its box
    # (benchmarks/grad_norm_spread.py); the reference value is 3.30754.  Hence 0.5 % on the total, 2 % per parameter.
    assert abs(total - mt["grad_sq_norm_total"]) <= 5e-3 * mt["grad_sq_norm_total"], (total, mt["grad_sq_norm_total"])
    for n, v in mt["grad_sq_norm_sample"].items():
        assert abs(gsq[n] - v) <= 2e-2 * max(v, 1e-12) + 1e-12, (n, gsq[n], v)
    # element-wise: whole gradient vectors of bias / norm parameters against the reference (make_golden_grads.py).
    # A squared norm cannot see a sign flip or a permutation (e.g. a bias gradient taken from column sums that belong
    # to another tensor); the direction can.  Same ~1 % arg-max-flip noise as above, hence 3 % of the vector's norm.
    gv = load("adapointr_grads.npz")
    named = dict(model.named_parameters())
    for n in gv.files:
        want = gv[n].astype(np.float64)
        got = named[n].grad.detach().double().cpu().numpy()
        assert got.shape == want.shape, n
        err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
        assert err < 3e-2, (n, err)
    return l_den.item(), l_rec.item()


def describe_eval_forward(device):
    """Diagnostics for the eval forward: per-sample max deviation, row-order agreement."""
    g = load("adapointr.npz")
    model, _ = build_adapointr(device)
    model.eval()
    with torch.no_grad():
        coarse, fine = model(t(g["partial"], device))
    c, w = coarse.cpu().numpy(), g["eval_coarse"]
    out = []
    for b in range(c.shape[0]):
        d = ((c[b][:, None, :] - w[b][None, :, :]) ** 2).sum(-1)
        out.append(dict(sample=b, same_index_max=float(np.abs(c[b] - w[b]).max()),
                        best_match_max=float(np.sqrt(d.min(1)).max()),
                        rows_out_of_order=int((d.argmin(1) != np.arange(c.shape[1])).sum()),
                        fine_max=float(np.abs(fine[b].cpu().numpy() - g["eval_fine"][b]).max())))
    return out


def build_img_model(device):
    from imgadapointr_amd.models import build_model_from_cfg
    from imgadapointr_amd.utils.config import cfg_from_yaml_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cwd = os.getcwd()
    os.chdir(root)
    try:
        cfg = cfg_from_yaml_file("cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml")
    finally:
        os.chdir(cwd)
    model = build_model_from_cfg(cfg.model)
    model.load_state_dict(seeded_state_dict(model.state_dict(), "imgadapointr"))
    return model.to(device), cfg


def check_img_model(device, tol=3e-4):
    """Image-fused model vs the reference fixture: eval forward, train forward, alpha-weighted loss."""
    from golden_util import synthetic_batch, synthetic_image
    g = load("imgadapointr.npz")
    mt = meta()
    model, _ = build_img_model(device)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == mt["img_state_dict"]
    assert sum(p.numel() for p in model.parameters()) == mt["img_n_params"]
    np.testing.assert_allclose(model.alpha_loss, g["alpha_loss"], rtol=1e-12)
    partial = t(synthetic_batch(1, 2048, 16384, seed=21, structured=True)[0], device)
    img = t(synthetic_image(1, seed=22), device)
    model.eval()
    with torch.no_grad():
        coarse, fine = model(partial, img)
    d_eval = compare_ranked(coarse.cpu().numpy(), fine[:, :4096].cpu().numpy(), g["eval_coarse"],
                            g["eval_fine_head"], tol)
    model.train()
    ret = model(partial, img, denoise_noise=t(g["train_noise"], device))
    d_train = (compare_ranked(ret[0].detach().cpu().numpy(), np.zeros((1, 0, 3), np.float32), g["pred_coarse"],
                              np.zeros((1, 0, 3), np.float32), tol)[0],
               maxdiff(ret[2][:, :512].detach().cpu().numpy(), g["denoised_fine_head"]))
    assert max(d_train) < tol, d_train
    gt = t(synthetic_batch(1, 2048, 16384, seed=23, structured=True)[1], device)
    l_den, l_rec = model.get_loss(ret, gt, 7)
    # model-level: a rank swap at the 512/513 boundary of the query selection moves the loss slightly
    assert abs(l_den.item() - float(g["loss_denoised"])) * 1000 < 2e-3
    assert abs(l_rec.item() - float(g["loss_recon"])) * 1000 < 2e-3
    # the reference raises IndexError from epoch 595 on; the table is clamped here
    assert model.coarse_loss_weight(600) == model.alpha_loss[-1]
    return d_eval, d_train


def check_seg_branch(device, tol=3e-4):
    """GDANet, normals and the image+segmentation fused model (eval) vs the reference fixture."""
    from golden_util import synthetic_batch, synthetic_image
    from imgadapointr_amd.models import build_model_from_cfg
    from imgadapointr_amd.models.segmentation.GDANet_ptseg import GDANet
    from imgadapointr_amd.models.segmentation.normals import estimate_pointcloud_normals
    from imgadapointr_amd.utils.config import cfg_from_yaml_file
    g = load("segadapointr.npz")
    mt = meta()
    gda = GDANet(50)
    gda.load_state_dict(seeded_state_dict(gda.state_dict(), "gdanet", std=0.1))
    gda = gda.to(device).eval()
    pts = t(synthetic_batch(1, 512, 512, seed=31, structured=False)[0], device)
    normals = estimate_pointcloud_normals(pts, 30, disambiguate_directions=False)
    d_n = maxdiff(normals[:, :96].cpu().numpy(), g["normals_head"])
    assert d_n < 1e-4, d_n
    assert float((normals.norm(dim=-1) - 1).abs().max()) < 1e-5
    cls = torch.zeros(1, 16, device=device)
    cls[0, 4] = 1.0
    with torch.no_grad():
        emb, logp = gda(pts.transpose(1, 2), normals, cls)
    d_g = (maxdiff(emb[:, :, :96].cpu().numpy(), g["gda_emb_head"]), maxdiff(logp[:, :96].cpu().numpy(), g["gda_logp_head"]))
    assert max(d_g) < tol, d_g

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cwd = os.getcwd()
    os.chdir(root)
    try:
        cfg = cfg_from_yaml_file("cfgs/SegImgPCN_models/ImgEncSegDecAdaPoinTrVariableLoss.yaml")
    finally:
        os.chdir(cwd)
    model = build_model_from_cfg(cfg.model)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == mt["seg_state_dict"]
    assert sum(p.numel() for p in model.parameters()) == mt["seg_n_params"]
    model.load_state_dict(seeded_state_dict(model.state_dict(), "imgsegadapointr"))
    model = model.to(device).eval()
    partial = t(synthetic_batch(1, 2048, 16384, seed=21, structured=True)[0], device)
    img = t(synthetic_image(1, seed=22), device)
    with torch.no_grad():
        coarse, fine = model(partial, img, cls)
    d_m = compare_ranked(coarse.cpu().numpy(), fine[:, :4096].cpu().numpy(), g["eval_coarse"],
                         g["eval_fine_head"], tol)
    return d_n, d_g, d_m


def check_chamfer_baselines(device, tol=1e-4):
    """PCN, FoldingNet and TopNet (token-major execution, split first layers) vs the reference fixture:
    state-dict contract, eval forward, train forward (batch statistics), both Chamfer-L2 losses, total gradient
    norm and the running mean the first BatchNorm accumulates."""
    import json
    from imgadapointr_amd.models import build_model_from_cfg
    from imgadapointr_amd.utils.config import cfg_from_yaml_file
    g = load("baselines.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = json.load(open(os.path.join(root, "tests", "golden", "baselines_meta.json")))["state_dict"]
    partial, gt = t(g["partial"], device), t(g["gt_head"], device)
    report = {}

    def rel(got, want):     # the seeded weights (std 0.1) give outputs of magnitude up to ~200: scale by the range
        return maxdiff(got.detach().cpu().numpy(), want) / max(1.0, float(np.abs(want).max()))
    for name in ("PCN", "FoldingNet", "TopNet"):
        cwd = os.getcwd()
        os.chdir(root)
        try:
            cfg = cfg_from_yaml_file(f"cfgs/PCN_models/{name}.yaml").model
        finally:
            os.chdir(cwd)
        if name == "TopNet":
            cfg.num_pred = 2048         # the reference cannot construct the 16384-leaf tree (see the generator)
        model = build_model_from_cfg(cfg)
        assert {k: list(v.shape) for k, v in model.state_dict().items()} == shapes[name], name
        model.load_state_dict(seeded_state_dict(model.state_dict(), f"baseline/{name}", std=0.1))
        model = model.to(device).eval()
        with torch.no_grad():
            coarse, fine = model(partial)
        d_eval = (rel(coarse[:, :512], g[f"{name}_eval_coarse"]), rel(fine[:, :1024], g[f"{name}_eval_fine_head"]))
        model.train()
        ret = model(partial)
        l0, l1 = model.get_loss(ret, gt)
        (l0 + l1).backward()
        d_train = rel(ret[1][:, :1024], g[f"{name}_train_fine_head"])
        want = g[f"{name}_losses"]
        d_loss = max(abs(l0.item() - want[0]) / abs(want[0]), abs(l1.item() - want[1]) / abs(want[1]))
        gsq = sum(float((p.grad.double() ** 2).sum()) for p in model.parameters() if p.grad is not None)
        d_g = abs(gsq - float(g[f"{name}_grad_sq_norm"])) / float(g[f"{name}_grad_sq_norm"])
        d_rm = maxdiff(model.first_conv[1].running_mean.cpu().numpy(), g[f"{name}_running_mean_first"])
        report[name] = (d_eval, d_train, d_loss, d_g, d_rm)
        # Train mode normalises with the statistics of an 8-cloud batch through up to 24 stacked BatchNorms (TopNet's
        # root layer sees 8 rows), which amplifies rounding: evaluated in float64 this network's gradient norm moves by
        # 2e-3 relative (4e-3 between two GEMM libraries), so that is the resolution of the gradient comparison.
        assert max(d_eval) < tol and d_train < 1e-3 and d_loss < 2e-4 and d_g < 1e-2 and d_rm < 1e-5, report
    return report


class _ChamferL2F64(torch.nn.Module):
    """ChamferDistanceL2 by brute force in the dtype of its inputs (the float64 referee of check_baseline_forms)."""

    def forward(self, a, b):
        d = (a.unsqueeze(2) - b.unsqueeze(1)).pow(2).sum(-1)
        return d.min(2)[0].mean() + d.min(1)[0].mean()


def check_baseline_forms(device):
    """Token-major / split-first-layer execution and the literal channel-first form of the reference, each judged
    against a FLOAT64 evaluation of the same modules (literal form, brute-force Chamfer): train-mode outputs, running
    statistics and the whole gradient (B = 8, small widths).

    Why a referee and not form vs form (round-2 review): these networks push 8-cloud batch statistics through up to
    24 stacked train-mode BatchNorms, and FoldingNet's gradient moves by 2e-4 when its float64 inputs move by one fp32
    ulp - so BOTH fp32 forms sit ~2e-2 from the float64 gradient (profiles/round3/baseline_forms_referee_v1.log:
    token-major 2.106e-2, literal 2.099e-2; PCN 9.1e-4 / 8.8e-4; TopNet 2.1e-4 / 2.1e-4) and their mutual distance
    (1.7e-3 on most boxes, 2e-2 on one: other library GEMM kernels) is a sample of that same rounding cloud, not a
    property of either form.  What can be asserted tightly is that the token-major form is no further from float64 than
    the reference's own form is."""
    from imgadapointr_amd.models import build_model_from_cfg, pointwise
    from imgadapointr_amd.utils.config import EasyDict
    from golden_util import synthetic_batch
    partial, gt = synthetic_batch(8, 256, 512, seed=51, structured=True)
    partial, gt = t(partial, device), t(gt, device)
    report = {}

    def run(cfg, token_major, dtype):
        pointwise.TOKEN_MAJOR = token_major
        try:
            model = build_model_from_cfg(EasyDict(cfg))
            model.load_state_dict(seeded_state_dict(model.state_dict(), "forms/" + cfg['NAME'], std=0.1))
            model = model.to(device).to(dtype).train()
            if dtype == torch.float64:
                model.loss_func = _ChamferL2F64()
            ret = model(partial.to(dtype))
            l0, l1 = model.get_loss(ret, gt.to(dtype))
            (l0 + l1).backward()
        finally:
            pointwise.TOKEN_MAJOR = True
        return (ret[1].detach().double(), {n: p.grad.double() for n, p in model.named_parameters()},
                {n: b.double() for n, b in model.named_buffers() if b.is_floating_point()})

    def grad_err(g, ref):
        return (sum(float((g[n] - ref[n]).pow(2).sum()) for n in ref) / sum(float(ref[n].pow(2).sum()) for n in ref)) ** 0.5

    for cfg in (dict(NAME='PCN', num_pred=1024, encoder_channel=1024),
                dict(NAME='FoldingNet', num_pred=1024, encoder_channel=256),
                dict(NAME='TopNet', node_feature=8, num_pred=2048, encoder_channel=248, nlevels=6)):
        o64, g64, b64 = run(cfg, False, torch.float64)
        # (1) the two forms are the SAME FUNCTION: in float64, where a near-tie of a max-pool arg-max would need a gap below
        # 1e-16, token-major and literal agree to accumulated double rounding in outputs, buffers and every gradient
        o64t, g64t, b64t = run(cfg, True, torch.float64)
        scale = float(o64.abs().max())
        same = dict(out=float((o64t - o64).abs().max()) / scale, grad=grad_err(g64t, g64),
                    buf=max(float((b64t[n] - b64[n]).abs().max() / (1 + b64[n].abs().max())) for n in b64))
        assert same["out"] < 1e-10 and same["grad"] < 1e-9 and same["buf"] < 1e-10, (cfg['NAME'], same)
        # (2) in fp32 each form is judged against float64.  Values: token-major no further than the literal form (1.5x: two
        # samples of one rounding cloud).  Gradients: only the conditioning envelope - the encoders pool with max over points,
        # and a near-tie there flips between implementations, boxes and ulp-level input perturbations alike, moving the
        # encoder's gradient in quantised jumps (measured on the CPU with 1e-7 input perturbations, FoldingNet: 3.9e-4,
        # 1.5e-3, 3.7e-3 depending on the seed - a float64 arm perturbed the same way jumps to 1.5e-3 too; on GPU boxes
        # 3.7e-3 or 2.1e-2 for either form).  An ordering of two such samples (round 3's `e_grad[0] <= 1.5 e_grad[1]`)
        # failed on the box where the literal form drew 3.7e-3 and the token-major one 2.1e-2 (pytest_gpu_r4_v1.log); with
        # (1) proving identity there is nothing left for it to show.
        (o1, g1, b1), (o2, g2, b2) = run(cfg, True, torch.float32), run(cfg, False, torch.float32)
        e_out = (float((o1 - o64).abs().max()) / scale, float((o2 - o64).abs().max()) / scale)
        e_grad = (grad_err(g1, g64), grad_err(g2, g64))
        e_buf = max(float((b1[n] - b64[n]).abs().max() / (1 + b64[n].abs().max())) for n in b64)
        report[cfg['NAME']] = dict(f64_form_vs_form=same, out_vs_f64=e_out, grad_vs_f64=e_grad, buf_vs_f64=e_buf,
                                   grad_form_vs_form=grad_err(g1, g2))
        assert e_out[0] <= 1.5 * e_out[1] + 1e-6, report
        assert max(e_out) < 5e-4 and max(e_grad) < 6e-2 and e_buf < 1e-5, report
    return report


def build_trainstep_model(case, device):
    """Model + config of a round-3 train-step case, seeded weights, segmentator frozen as tools/runner.py:55-66 does."""
    from golden_util import TRAINSTEP_CASES, trainstep_state_dict
    from imgadapointr_amd.models import build_model_from_cfg
    from imgadapointr_amd.utils.config import cfg_from_yaml_file
    spec = TRAINSTEP_CASES[case]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cwd = os.getcwd()
    os.chdir(root)
    try:
        cfg = cfg_from_yaml_file(spec["yaml"])
    finally:
        os.chdir(cwd)
    model = build_model_from_cfg(cfg.model)
    model.load_state_dict(trainstep_state_dict(model.state_dict(), case))
    if spec["frozen_segmentator"]:
        for p in model.base_model.segmentator.parameters():
            p.requires_grad = False
    return model.to(device), cfg, spec


class _FixedDropout(torch.nn.Module):
    """nn.Dropout(p) with the keep-mask given instead of drawn."""

    def __init__(self, keep, p):
        super().__init__()
        self.keep, self.p = keep, p

    def forward(self, x):
        return x * self.keep * (1.0 / (1.0 - self.p)) if self.training else x


class _Pinned:
    """Library choices that are free to differ from run to run, pinned for the length of a parity run: MIOpen's
    per-process search (`benchmark`) off, its deterministic algorithms on, ATen's deterministic scatter forms on."""

    def __enter__(self):
        b = torch.backends.cudnn
        self.saved = (b.deterministic, b.benchmark, torch.are_deterministic_algorithms_enabled(),
                      torch.is_deterministic_algorithms_warn_only_enabled())
        b.deterministic, b.benchmark = True, False
        # ATen's scatter / index_add / gather-backward kernels add with float atomics in arrival order unless told otherwise
        # (GDANet's feature-space neighbourhoods and the normals go through them); warn_only: an op without a
        # deterministic form is reported, not refused
        torch.use_deterministic_algorithms(True, warn_only=True)
        return self

    def __exit__(self, *exc):
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = self.saved[:2]
        torch.use_deterministic_algorithms(self.saved[2], warn_only=self.saved[3])


def prepare_trainstep(case, device, replay_scores=True):
    """Model, production `TrainStep`, inputs and fixture of a train-step case with every recorded RNG consumer and chaotic
    decision of the reference's forward replayed: the denoise jitter (passed as `denoise_noise`), the two Dropout(0.4)
    keep-masks of the train-mode GDANet (GDANet_ptseg.py:59-61) and — round 4 — the query ranking
    (AdaPoinTr.py:119-121: the 512-of-768 selection; two scores closer than rounding swap places between implementations).
    The reference's order is replayed by a forward hook on the `query_ranking` module, whose parameters receive no gradient in
    either implementation (argsort is not differentiable), so nothing on the product side knows about it."""
    from golden_util import trainstep_inputs
    from imgadapointr_amd.tools import builder
    from imgadapointr_amd.tools.runner import TrainStep
    g = load(f"trainstep_{case}.npz")
    with open(os.path.join(GOLD, f"trainstep_{case}.json")) as fh:
        mt = json.load(fh)
    model, cfg, spec = build_trainstep_model(case, device)
    model.train()
    step = TrainStep(model, builder.build_optimizer(model, cfg), getattr(cfg, "grad_norm_clip", 10), 1)
    partial, gt, img, cls = trainstep_inputs(case)
    extra = []
    if img is not None:
        extra.append(t(img, device))
    elif spec["takes_img"]:
        extra.append(None)
    if cls is not None:
        extra.append(t(cls, device))
    for i, name in enumerate(("dp1", "dp2")):
        if f"dropout_mask_{i}" in g.files:
            shape = tuple(int(x) for x in g[f"dropout_mask_{i}_shape"])
            mask = np.unpackbits(g[f"dropout_mask_{i}"])[:int(np.prod(shape))].reshape(shape).astype(np.float32)
            setattr(model.base_model.segmentator, name, _FixedDropout(t(mask, device), 0.4))
    if replay_scores:
        # a tie-free surrogate with exactly the reference's order: the Sigmoid of query_ranking leaves EXACT ties between
        # different proposals (img fixture: smallest gap inside the top 512 is 0.0), and how an argsort orders ties is the
        # implementation's business (CPU and GPU sorts differ)
        order = torch.from_numpy(g["query_order"].astype(np.int64))
        scores = torch.empty(order.shape, dtype=torch.float32)
        scores.scatter_(1, order, (1.0 - torch.arange(order.size(1), dtype=torch.float32) / 1024.0).expand_as(scores))
        scores = scores.unsqueeze(-1).to(device)
        model.base_model.query_ranking.register_forward_hook(lambda m, a, o: scores.to(o.dtype))

    def run():
        return step(t(partial, device), t(gt, device), tuple(extra), epoch=spec["epoch"],
                    denoise_noise=t(g["train_noise"], device))
    return model, cfg, spec, step, run, g, mt


def _rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def check_trainstep(case, device):
    """ONE full `TrainStep` (forward, get_loss, backward, clip + AdamW — the production step object of
    tools/runner.py, fast paths as shipped) of a fused model family at B = 2 against the reference fixture
    tests/golden/trainstep_<case>.npz: train outputs, both losses (CD-L1 x1000 within the north_star's 1e-3), the set
    of parameters without gradient, total and per-tensor gradient norms, whole gradient vectors, parameters after the
    update, and the frozen segmentator's running statistics (its BatchNorms stay in train mode, runner.py:115).

    EVERY deviation is collected and reported in one assertion, together with what localises it: the outputs of the two
    library-backed branches (ResNet18 / GDANet: MIOpen convolutions on the GPU) against the reference's, and the tensors
    whose |g|^2 moved most — round 3's one failing run (total |g|^2 2.1 % low, everything else in bounds) could not say
    what had moved."""
    with _Pinned():
        return _check_trainstep(case, device)


def _check_trainstep(case, device):
    model, cfg, spec, step, run, g, mt = prepare_trainstep(case, device)
    named = dict(model.named_parameters())
    assert sum(1 for p in named.values() if p.requires_grad) == mt["n_trainable"]
    before = {n[len("after/"):]: named[n[len("after/"):]].detach().clone() for n in g.files if n.startswith("after/")}
    grads, outs, branch = {}, [], {}
    step.before_update = lambda: grads.update({n: p.grad.detach().clone() for n, p in named.items() if p.grad is not None})
    hooks = [model.register_forward_hook(lambda m, a, o: outs.append(o))]
    if hasattr(model.base_model, "im_encoder"):
        hooks.append(model.base_model.im_encoder.register_forward_hook(
            lambda m, a, o: branch.__setitem__("img", o.detach())))
    if hasattr(model.base_model, "segmentator"):
        hooks.append(model.base_model.segmentator.register_forward_hook(
            lambda m, a, o: branch.__setitem__("seg", o[0].detach())))
    l_den, l_rec = run()
    for h in hooks:
        h.remove()
    ret = outs[0]
    B = ret[0].shape[0]
    problems, report = [], {}

    def check(name, value, limit):
        report[name] = value
        if not value < limit:
            problems.append(f"{name} = {value:.3e} (limit {limit:.1e})")

    # the library-backed branches first: they are upstream of everything below
    if "img" in branch:
        check("img_tokens_rel", _rel(branch["img"].reshape(B, -1)[:, :4096].cpu().numpy(), g["img_tokens_head"]), 1e-4)
        check("img_tokens_sq_norm_rel", abs(float((branch["img"].double() ** 2).sum()) / float(g["img_tokens_sq_norm"]) - 1), 1e-4)
    if "seg" in branch:
        # GDANet in train mode over B = 2 (batch statistics, feature-space top-k): compared for the record, bounded loosely
        check("seg_emb_rel", _rel(branch["seg"].reshape(B, -1)[:, :4096].cpu().numpy(), g["seg_emb_head"]), 5e-2)
    tol = 3e-4
    check("out_denoised_coarse", maxdiff(ret[1].detach().cpu().numpy(), g["denoised_coarse"]), tol)
    check("out_denoised_fine", maxdiff(ret[2][:, :512].detach().cpu().numpy(), g["denoised_fine_head"]), tol)
    # with the scores replayed the queries keep the reference's order: rows are compared in place
    check("out_pred_coarse", maxdiff(ret[0].detach().cpu().numpy(), g["pred_coarse"]), tol)
    check("out_pred_fine", maxdiff(ret[3][:, :2048].detach().cpu().numpy(), g["pred_fine_head"]), tol)
    # losses x 1000 (the logged unit): the north_star's 1e-3 for every case
    check("loss_denoised_x1000", abs(l_den.item() - float(g["loss_denoised"])) * 1000, 1e-3)
    check("loss_recon_x1000", abs(l_rec.item() - float(g["loss_recon"])) * 1000, 1e-3)
    no_grad = sorted(n for n, p in named.items() if p.requires_grad and n not in grads)
    if no_grad != mt["no_grad_params"]:
        problems.append(f"parameters without gradient: {sorted(set(no_grad) ^ set(mt['no_grad_params']))}")
    if any(n in grads for n, p in named.items() if not p.requires_grad):
        problems.append("a frozen parameter received a gradient")
    gsq = {n: float((v.double() ** 2).sum()) for n, v in grads.items()}
    total, want_total = sum(gsq.values()), float(g["grad_sq_norm_total"])
    # Resolution of the gradient comparison = what rounding alone does to this fixture, measured by running the step
    # with the GEMM / attention / edge-conv kernels exchanged for the library's and the literal forms
    # (benchmarks/trainstep_spread.py -> profiles/round3/trainstep_spread_v1.log) and, round 4, with the branch outputs
    # perturbed on the CPU (benchmarks/trainstep_sensitivity.py -> profiles/round4/trainstep_sensitivity_v1.log).
    # img / segenc: arms agree to 4e-3 in the gradient vector and 1.4e-3 in |g|^2 - the AdaPoinTr resolution (arg-max
    # flips at near-ties); relative perturbations of the image tokens from 1e-6 to 1e-3 all move |g|^2 by <= 1e-3, and it
    # takes 1e-2 (bf16-level precision) to move it by the 2 % of round 3's failing run - which is why the image tokens are
    # compared first.  seg: the arms differ from EACH OTHER by 2-5 % in the vector and 0.5 % in |g|^2 (GDANet in train
    # mode over B = 2: feature-space top-k, top-M sharp/gentle selection and a hard distance mask after the predicted
    # coarse cloud; an fp32-ulp perturbation of the cloud handed to GDANet moves |g|^2 by 2.4e-3), so that is all this
    # fixture can resolve; outputs and losses stay at 3e-4 / 1e-3 for every case on every device.
    loose = case == "seg"
    tol_total, tol_vec = (3e-2, 8e-2) if loose else (5e-3, 3e-2)
    check("grad_sq_total_rel", abs(total - want_total) / want_total, tol_total)
    want_all = mt["grad_sq_norm_all"]
    # per tensor: deviation of |g_n|^2 as a share of its own size (tensors above 1e-9 of the total) ...
    dev = {n: abs(gsq.get(n, 0.0) - v) / v for n, v in want_all.items() if v > 1e-9 * want_total}
    worst = sorted(dev, key=dev.get, reverse=True)[:6]
    report["grad_sq_worst_tensors"] = {n: (gsq.get(n, 0.0), want_all[n]) for n in worst}
    check("grad_sq_per_tensor_rel", dev[worst[0]], 5 * tol_vec)
    # ... and as a share of the total: which tensors carry a deviation of the total
    share = {n: (gsq.get(n, 0.0) - v) / want_total for n, v in want_all.items()}
    carriers = sorted(share, key=lambda n: abs(share[n]), reverse=True)[:6]
    report["grad_sq_total_carriers"] = {n: share[n] for n in carriers}
    d_norm = max(abs(gsq[n] - v) / max(v, 1e-12) for n, v in mt["grad_sq_norm_sample"].items() if v > 1e-9 * want_total)
    check("grad_sq_sample_rel", d_norm, tol_vec)
    d_vec = {}
    for key in g.files:
        if key.startswith("grad/"):
            n = key[len("grad/"):]
            want = g[key].astype(np.float64)
            got = grads[n].double().cpu().numpy()
            if np.linalg.norm(want) < 1e-4 * np.sqrt(want_total):
                # e.g. a k_map bias: softmax is invariant to a constant added to every key's logit, so this gradient is
                # zero in exact arithmetic and pure rounding noise in both implementations; only its size is checked
                check("grad_null/" + n, float(np.linalg.norm(got)), 1e-3 * np.sqrt(want_total))
                continue
            d_vec[n] = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    assert len(d_vec) >= 12
    worst_vec = max(d_vec, key=d_vec.get)
    report["grad_vec_worst"] = worst_vec
    check("grad_vec_rel", d_vec[worst_vec], tol_vec)
    check("grad_norm_before_clip_rel", abs(float(np.sqrt(total)) / float(g["grad_norm_before_clip"]) - 1), tol_total)
    # after the update: AdamW's first step moves every element by ~lr * sign(g) (lr 1e-4), so an element whose gradient
    # is at rounding level may land 2e-4 away; everything else must agree to a small fraction of one step
    lr = float(cfg.optimizer.kwargs.lr)
    moved, far, n_el = 0.0, 0, 0
    for n, b in before.items():
        want = g["after/" + n]
        got = named[n].detach().reshape(-1)[:4096].cpu().numpy()
        was = b.reshape(-1)[:4096].cpu().numpy()
        if np.abs(want - was).max() < 0.2 * lr:       # a k_map bias again: gradient at rounding level, below Adam's eps
            check("after_null/" + n, float(np.abs(got - was).max()), 1.1 * lr)
            continue
        diff = np.abs(got - want)
        far += int((diff > 0.05 * lr).sum())
        n_el += diff.size
        check("after_max/" + n, float(diff.max()), 2.2 * lr + 1e-12)
        moved = max(moved, float(np.abs(got - was).max()))
    # a gradient vector that agrees to 1-2 % (arg-max flips, above) has about that share of elements near zero whose
    # sign, and with it the whole first step, may differ
    check("after_far_frac", far / n_el, 0.12 if loose else 0.06)
    check("after_step_size_low", 0.5 * lr - moved, 0.0)
    check("after_step_size_high", moved - 1.5 * lr, 0.0)
    if "seg_first_running_mean" in g.files:
        bn = next(m for m in model.base_model.segmentator.modules() if isinstance(m, torch.nn.BatchNorm2d))
        check("seg_running_mean", maxdiff(bn.running_mean.cpu().numpy(), g["seg_first_running_mean"]), 1e-5)
        if not all(p.grad is None for p in model.base_model.segmentator.parameters()):
            problems.append("the frozen segmentator holds a gradient")
    assert not problems, "train step %s on %s deviates from the reference fixture:\n  %s\nfull report: %s" % (
        case, device, "\n  ".join(problems), json.dumps(report, indent=1, sort_keys=True, default=str))
    return report


def check_trainstep_reproducible(case, device):
    """The same `TrainStep` of a fused family twice in ONE process from the same weights: every gradient outside the
    library-backed branches (ResNet18 / GDANet run on MIOpen, whose kernels this build does not own) must come out
    bit-identical - the padded-head `multihead_attention`, add + LayerNorm and fusion layers run on libptops kernels, and a
    race there would show here, not as "MIOpen picked another kernel".  Returns (number of tensors compared, the names of
    library-branch tensors that differ between the runs - informational)."""
    with _Pinned():
        got, first = [], {}
        for r in range(2):
            torch.manual_seed(0)
            model, cfg, spec, step, run, g, mt = prepare_trainstep(case, device)
            grads, same = {}, {}

            def pin(key):
                # run 2 continues from run 1's branch output (written into the output in place, so the branch's own
                # backward is untouched): what follows the branch then sees identical bits whatever MIOpen did
                def hook(m, a, o):
                    out = o[0] if isinstance(o, tuple) else o
                    if r == 0:
                        first[key] = out.detach().clone()
                    else:
                        same[key] = bool(torch.equal(out.detach(), first[key]))
                        out.data.copy_(first[key])
                return hook
            for key in ("im_encoder", "segmentator"):
                if hasattr(model.base_model, key):
                    getattr(model.base_model, key).register_forward_hook(pin(key))
            step.before_update = lambda: grads.update(
                {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
            losses = run()
            got.append((grads, [x.item() for x in losses], same))
            del model, step
    (g1, l1, _), (g2, l2, branch_same) = got
    assert g1.keys() == g2.keys()
    lib = lambda n: ".im_encoder." in n or ".segmentator." in n        # noqa: E731
    differ = sorted(n for n in g1 if not torch.equal(g1[n], g2[n]))
    ours = [n for n in differ if not lib(n)]
    assert l1 == l2 and not ours, (l1, l2, "branch outputs bit-identical between the runs: %s" % branch_same,
                                   {n: float((g1[n] - g2[n]).abs().max() / g1[n].abs().max()) for n in ours[:8]})
    return sum(1 for n in g1 if not lib(n)), [n for n in differ if lib(n)]


def check_multi_view(device, views=3):
    """The multi-view batch ([B,V,3,224,224] renders, README.md:80-85 protocol; point branch evaluated once per
    object) against V separate single-view forwards: eval outputs, and in train mode the loss of the folded batch
    against the mean of the per-view losses (same jitter per view; the ResNet BatchNorms are put in eval mode for
    this comparison because batch statistics over B*V images differ from those over B by construction)."""
    from golden_util import synthetic_batch, synthetic_image
    model, _ = build_img_model(device)
    B = 2
    partial, gt = synthetic_batch(B, 2048, 16384, seed=61, structured=True)
    partial, gt = t(partial, device), t(gt, device)
    imgs = t(synthetic_image(B * views, seed=62).reshape(B, views, 3, 224, 224), device)
    model.eval()
    with torch.no_grad():
        coarse, fine = model(partial, imgs)
        assert coarse.shape == (B * views, 512, 3) and fine.shape == (B * views, 16384, 3)
        d, swapped = 0.0, 0
        for v in range(views):
            c1, f1 = model(partial, imgs[:, v].contiguous())
            # the 512 queries are the top-ranked of 768 proposals: two scores closer than fp32 rounding may swap places
            # between a batch of B*V and a batch of B (other GEMM kernels), so rows are matched before comparing
            dc, df, sw = compare_ranked(coarse[v::views].cpu().numpy(), fine[v::views, :2048].cpu().numpy(),
                                        c1.cpu().numpy(), f1[:, :2048].cpu().numpy(), 2e-5)
            d, swapped = max(d, dc, df), swapped + sw
    assert d < 2e-5 and swapped <= 8, (d, swapped)
    # train mode: losses and the gradient that reaches the shared point branch
    model.train()
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.eval()
    noise = (torch.randn(B, 64, 3, generator=torch.Generator().manual_seed(5)) * 0.01).clamp_(-0.05, 0.05).to(device)
    ret = model(partial, imgs, denoise_noise=noise.repeat_interleave(views, 0))
    l_den, l_rec = model.get_loss(ret, gt, 7)
    (l_den + l_rec).backward()
    w = model.base_model.grouper.input_trans.weight
    g_folded = w.grad.detach().clone()
    model.zero_grad(set_to_none=True)
    acc = [0.0, 0.0]
    for v in range(views):
        r = model(partial, imgs[:, v].contiguous(), denoise_noise=noise)
        a, b = model.get_loss(r, gt, 7)
        ((a + b) / views).backward()
        acc[0] += a.item() / views
        acc[1] += b.item() / views
    d_loss = (abs(l_den.item() - acc[0]), abs(l_rec.item() - acc[1]))
    d_grad = float((g_folded - w.grad).norm() / w.grad.norm())
    # CPU: identical arithmetic in both forms.  GPU: a batch of B*V rows runs other GEMM kernels than a batch of B, so
    # rounding differs (and with it the odd arg-max / rank decision): losses to 5e-5, the shared branch's gradient to 3 %
    on_cpu = str(device) == "cpu"
    assert max(d_loss) < (2e-6 if on_cpu else 5e-5) and d_grad < (2e-3 if on_cpu else 3e-2), (d_loss, d_grad)
    return d, d_loss, d_grad


def check_validate(device):
    """`tools.runner.validate` + `Metrics.get` on fixed predictions against the reference's own `validate()` run on the same
    stub model and loader (tests/golden/metrics_validate.json, generator make_golden_round3.py::metrics_fixture): the four
    losses, F-Score / CDL1 / CDL2 per sample and per taxonomy, the printed TEST RESULTS table, the returned Metrics object
    and its ordering rule.  F-Score: the reference thresholds open3d's float64 nearest-neighbour distances, this build the
    square roots of the Chamfer kernel's float32 squared distances; the fixture counts the distances within 2e-6 of the
    threshold (3 of 24576), which bounds the possible disagreement."""
    import re
    from types import SimpleNamespace
    from golden_util import metrics_cases
    from imgadapointr_amd.extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
    from imgadapointr_amd.tools import runner
    from imgadapointr_amd.utils.config import EasyDict
    from imgadapointr_amd.utils.metrics import Metrics
    with open(os.path.join(GOLD, "metrics_validate.json")) as fh:
        want = json.load(fh)
    cases = metrics_cases()

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.calls = 0
            self.anchor = torch.nn.Parameter(torch.zeros(1))        # validate() reads the device off the parameters

        def forward(self, partial):
            c = cases[self.calls]
            self.calls += 1
            return t(c["coarse"], device)[None], t(c["dense"], device)[None]

    loader = [((c["taxonomy"],), (c["model"],), (torch.from_numpy(c["partial"])[None], torch.from_numpy(c["gt"])[None]))
              for c in cases]
    cfg = EasyDict(dataset=EasyDict(val=EasyDict(_base_=EasyDict(N_POINTS=2048, NAME="PCN"))), consider_metric="CDL1")
    args = SimpleNamespace(distributed=False, graph_eval=False)
    lines = []
    saved = runner.print_log
    runner.print_log = lambda msg, logger=None: lines.append(str(msg))
    try:
        metrics, l1, l2 = runner.validate(Stub().to(device), loader, 3, ChamferDistanceL1(), ChamferDistanceL2(), None, args, cfg)
    finally:
        runner.print_log = saved
    assert Metrics.names() == want["names"] and runner._synset_names() == want["synset_names"]
    slack = want["distances_within_2e-6_of_the_threshold"] / 2048.0 + 1e-6       # what threshold flips could move an F-Score by
    got = metrics.state_dict()
    for k, v in want["metrics"].items():
        assert abs(got[k] - v) <= (slack if k == "F-Score" else 1e-5 * max(1.0, abs(v))), (k, got[k], v)
    assert np.allclose(l1, want["loss_L1"], rtol=1e-5, atol=1e-6) and np.allclose(l2, want["loss_L2"], rtol=1e-5, atol=1e-7)
    for c, ref_vals in zip(cases, want["per_sample"]):
        vals = [float(v) for v in Metrics.get(t(c["dense"], device)[None], t(c["gt"], device)[None])]
        assert abs(vals[0] - ref_vals[0]) <= slack and np.allclose(vals[1:], ref_vals[1:], rtol=1e-5, atol=1e-6), (vals, ref_vals)
    # the log: same lines in the same order; numbers inside them to the printed precision (+ the F-Score slack)
    number = re.compile(r"-?\d+\.\d+")
    assert len(lines) == len(want["lines"]), (lines, want["lines"])
    for a, b in zip(lines, want["lines"]):
        assert number.sub("#", a) == number.sub("#", b), (a, b)
        for x, y in zip(number.findall(a), number.findall(b)):
            assert abs(float(x) - float(y)) <= 1.1e-3 + slack, (a, b)
    assert [Metrics("CDL1", {"CDL1": 1.0}).better_than(Metrics("CDL1", {"CDL1": 2.0})),
            Metrics("F-Score", {"F-Score": 0.3}).better_than(Metrics("F-Score", {"F-Score": 0.2})),
            Metrics("CDL2", {"CDL2": 1.0}).better_than(None)] == want["better_than"]
    return got


def check_inference(device, tmp_path):
    """`tools/inference.py` as a command line (--pc_root mode, reference inference.py:48-122) on the jittered demo scan
    against the reference's fine.npy (tests/golden/inference.npz): checkpoint loading through `builder.load_model`, the
    reader, UpSamplePoints with the same numpy RNG state, the eval forward, the output path.  Query groups (32 points each)
    are matched by their ranking before comparing, as in `check_eval_forward`."""
    from golden_util import inference_input, write_ascii_pcd
    from imgadapointr_amd.tools import inference
    g = load("inference.npz")
    raw = load("demo_data.npz")["raw/02691156"]
    model, _ = build_adapointr("cpu")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ckpt = os.path.join(str(tmp_path), "ckpt.pth")
    torch.save({"base_model": {"module." + k: v for k, v in model.state_dict().items()}}, ckpt)   # a DataParallel checkpoint
    os.makedirs(os.path.join(str(tmp_path), "in"))
    write_ascii_pcd(os.path.join(str(tmp_path), "in", "plane.pcd"), inference_input(raw))
    np.random.seed(77)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        inference.main(["cfgs/PCN_models/AdaPoinTr.yaml", ckpt, "--pc_root", os.path.join(str(tmp_path), "in"),
                        "--out_pc_root", os.path.join(str(tmp_path), "out"), "--device", str(device)])
    finally:
        os.chdir(cwd)
    fine = np.load(os.path.join(str(tmp_path), "out", "plane", "fine.npy"))
    want = g["fine"]
    assert fine.shape == want.shape == (16384, 3) and fine.dtype == np.float32
    got_g, want_g = fine.reshape(512, 32, 3), want.reshape(512, 32, 3)
    p = _match_rows(got_g.mean(1), want_g.mean(1))
    swapped = int((p != np.arange(512)).sum())
    d = maxdiff(got_g, want_g[p])
    assert d < 2e-4 and swapped <= 8, (d, swapped)
    return d, swapped
