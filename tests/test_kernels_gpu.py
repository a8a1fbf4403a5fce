"""Parity of the HIP kernels (through the C ABI) against the CPU oracle on seeded inputs.

Indices and distances must be BIT-EXACT; scatter-add backward passes (floating-point
accumulation order differs between the GPU's atomics and the oracle's loops) are compared
with a tight tolerance written next to each check.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cloud(rng, B, n, kind="uniform"):
    if kind == "uniform":
        return (rng.random((B, n, 3), dtype=np.float32) - 0.5)
    if kind == "shell":  # points on a random-radius sphere shell, 25 % trailing zero rows
        v = rng.standard_normal((B, n, 3)).astype(np.float32)
        v /= np.linalg.norm(v, axis=-1, keepdims=True) + 1e-9
        v *= (0.3 + 0.2 * rng.random((B, n, 1), dtype=np.float32))
        v[:, n - n // 4:, :] = 0.0
        return v.astype(np.float32)
    if kind == "lattice":  # integer grid: masses of exactly equal distances between DISTINCT points
        g = rng.integers(0, 12, size=(B, n, 3)).astype(np.float32) * np.float32(0.0625) - np.float32(0.375)
        return g
    if kind == "clusters":  # a few tight blobs far apart (pruning-hostile for the far cloud)
        centres = rng.random((B, 6, 3), dtype=np.float32) - 0.5
        pick = rng.integers(0, 6, size=(B, n))
        pts = np.take_along_axis(centres, pick[..., None].repeat(3, -1), axis=1)
        return (pts + rng.standard_normal((B, n, 3)).astype(np.float32) * np.float32(0.004)).astype(np.float32)
    if kind == "plane":  # every point on one x: the sweep can prune nothing
        v = rng.random((B, n, 3), dtype=np.float32) - 0.5
        v[..., 0] = np.float32(0.25)
        return v
    if kind == "dups":  # heavy duplication -> exact distance ties
        base = (rng.random((B, max(n // 8, 1), 3), dtype=np.float32) - 0.5)
        idx = rng.integers(0, base.shape[1], size=(B, n))
        return np.take_along_axis(base, idx[..., None].repeat(3, -1), axis=1).astype(np.float32)
    raise ValueError(kind)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


CHAMFER_CASES = [
    (2, 64, 128, "uniform"), (4, 2048, 2048, "uniform"), (3, 512, 16384, "uniform"),
    (2, 16384, 16384, "uniform"), (1, 1, 1, "uniform"), (1, 1, 777, "uniform"), (2, 300, 17, "uniform"),
    (2, 1000, 999, "dups"), (2, 2048, 4096, "shell"), (1, 5000, 3, "uniform"), (32, 512, 2048, "uniform"),
    # the pruned sweep path (both clouds in [1024, 16384]): ties, clusters, degenerate x, ragged sizes
    (2, 4096, 4096, "lattice"), (2, 4100, 5000, "dups"), (2, 8192, 4099, "clusters"), (2, 4096, 16384, "plane"),
    (3, 4097, 4111, "uniform"), (1, 16384, 16384, "shell"), (2, 6000, 5000, "lattice"), (1, 16383, 4096, "clusters"),
]


@pytest.mark.parametrize("B,n,m,kind", CHAMFER_CASES)
def test_chamfer_forward_bit_exact(native, ref, B, n, m, kind):
    rng = np.random.default_rng(B * 1000003 + n * 101 + m)
    a, b = _cloud(rng, B, n, kind), _cloud(rng, B, m, kind)
    r1, r2, j1, j2 = ref.chamfer_forward(a, b)
    algos = [native.CHAMFER_BRUTE, native.CHAMFER_AUTO, native.CHAMFER_MFMA,
             (native.CHAMFER_MFMA, native.CHAMFER_BRUTE), (native.CHAMFER_BRUTE, native.CHAMFER_MFMA)]
    if native.load().ptops_chamfer_sweep_available(B, n, m):
        algos += [native.CHAMFER_SWEEP, (native.CHAMFER_SWEEP, native.CHAMFER_BRUTE),
                  (native.CHAMFER_MFMA, native.CHAMFER_SWEEP)]
    for algo in algos:   # every algorithm must reproduce the oracle bit for bit
        d1, d2, i1, i2 = native.chamfer_forward(_t(a), _t(b), algorithm=algo)
        assert np.array_equal(d1.cpu().numpy().view(np.uint32), r1.view(np.uint32)), algo
        assert np.array_equal(d2.cpu().numpy().view(np.uint32), r2.view(np.uint32)), algo
        assert np.array_equal(i1.cpu().numpy(), j1), algo
        assert np.array_equal(i2.cpu().numpy(), j2), algo


def test_chamfer_mixed_kinds(native, ref):
    """Queries and candidates of different structure (uniform gt vs clustered / lattice predictions)."""
    rng = np.random.default_rng(99)
    for ka, kb, n, m in (("uniform", "clusters", 16384, 16384), ("lattice", "uniform", 4096, 8192),
                         ("clusters", "lattice", 5000, 4500)):
        a, b = _cloud(rng, 2, n, ka), _cloud(rng, 2, m, kb)
        want = ref.chamfer_forward(a, b)
        for algo in (native.CHAMFER_SWEEP, native.CHAMFER_BRUTE, native.CHAMFER_MFMA):
            got = native.chamfer_forward(_t(a), _t(b), algorithm=algo)
            for g, w in zip(got, want):
                assert np.array_equal(g.cpu().numpy(), w), (ka, kb, algo)


@pytest.mark.parametrize("kind,scale,shift", [("uniform", 1.0, 0.0), ("uniform", 1000.0, 0.0), ("uniform", 1e-3, 0.0),
                                              ("uniform", 1.0, 100.0), ("clusters", 1.0, 3.0), ("dups", 50.0, -7.0),
                                              ("lattice", 1.0, 0.0)])
def test_chamfer_mfma_filter_error_bound(native, ref, kind, scale, shift):
    """The MFMA filter's error bound scales with (|c|max + |q|)^2: clouds far from the origin, tiny and huge clouds,
    and masses of exact ties (every lane falls back to its exact scan) must all stay bit-exact."""
    rng = np.random.default_rng(int(scale * 7 + shift * 13) + len(kind))
    B, n, m = 2, 3000, 2100
    a = (_cloud(rng, B, n, kind) * np.float32(scale) + np.float32(shift)).astype(np.float32)
    b = (_cloud(rng, B, m, kind) * np.float32(scale) + np.float32(shift)).astype(np.float32)
    want = ref.chamfer_forward(a, b)
    got = native.chamfer_forward(_t(a), _t(b), algorithm=native.CHAMFER_MFMA)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy().view(np.uint32), w.view(np.uint32)), (kind, scale, shift)


def test_chamfer_adaptive_policy(native, monkeypatch):
    """The adaptive choice never changes results; a direction whose sweep cannot prune is demoted to the MFMA filter,
    and an MFMA filter drowning in exact ties is demoted to brute force."""
    key = (0, 4, 8192, 8192)
    pol = native._chamfer_policy
    pol.state.clear()
    monkeypatch.setattr(pol, "mfma_min_pairs", 0)                     # small clouds here: let the MFMA rung take part
    gt = torch.rand(4, 8192, 3, device="cuda") - 0.5
    blob = (torch.randn(4, 8192, 3, device="cuda") * 1e-3)           # the whole cloud in one tiny cluster
    ref_out = native.chamfer_forward(blob, gt, algorithm=native.CHAMFER_BRUTE)
    level = [0, 0]
    seen = set()
    for _ in range(6):
        out = native.chamfer_forward(blob, gt)
        seen.add(native.chamfer_algorithm_log[(4, 8192, 8192)])
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, ref_out))
        for d in range(2):
            level[d] = max(level[d], pol.state[key]["level"][d])
    # uniform queries against one cluster (direction 2) cannot prune: MFMA filter; cluster queries against a uniform
    # cloud can: the sweep stays
    assert level == [0, 1], level
    assert (native.CHAMFER_SWEEP, native.CHAMFER_MFMA) in seen, seen
    pol.state.clear()
    near = gt + torch.randn_like(gt) * 0.005                          # prediction close to the target
    for _ in range(6):
        native.chamfer_forward(near, gt)
        torch.cuda.synchronize()
    assert pol.state[key]["level"] == [0, 0]                          # both directions keep sweeping
    # not sweep-eligible (2048 points): MFMA by default; a lattice (masses of exact ties) sends it to brute force
    pol.state.clear()
    lat = (torch.randint(0, 6, (4, 2048, 3), device="cuda").float() * 0.125 - 0.3)
    lat2 = (torch.randint(0, 6, (4, 2048, 3), device="cuda").float() * 0.125 - 0.3)
    want = native.chamfer_forward(lat, lat2, algorithm=native.CHAMFER_BRUTE)
    used = []
    for _ in range(6):
        out = native.chamfer_forward(lat, lat2)
        used.append(native.chamfer_algorithm_log[(4, 2048, 2048)])
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, want))
    assert used[0] == (native.CHAMFER_MFMA, native.CHAMFER_MFMA) and used[-1] == (native.CHAMFER_BRUTE, native.CHAMFER_BRUTE), used
    pol.state.clear()


def test_chamfer_identity(native):
    x = torch.rand(2, 1000, 3, device="cuda") - 0.5
    d1, d2, i1, i2 = native.chamfer_forward(x, x.clone())
    assert float(d1.abs().max()) == 0.0 and float(d2.abs().max()) == 0.0
    ar = torch.arange(1000, device="cuda", dtype=torch.int32).expand(2, -1)
    assert torch.equal(i1, ar) and torch.equal(i2, ar)


@pytest.mark.parametrize("B,n,m,kind", [(2, 64, 128, "uniform"), (4, 2048, 2048, "uniform"),
                                        (2, 512, 16384, "uniform"), (2, 1000, 999, "dups")])
def test_chamfer_backward(native, ref, B, n, m, kind):
    rng = np.random.default_rng(7 + n + m)
    a, b = _cloud(rng, B, n, kind), _cloud(rng, B, m, kind)
    _, _, j1, j2 = ref.chamfer_forward(a, b)
    g1 = rng.standard_normal((B, n)).astype(np.float32)
    g2 = rng.standard_normal((B, m)).astype(np.float32)
    ra, rb = ref.chamfer_backward(a, b, j1, j2, g1, g2)
    # the reference's form (chamfer.cu:193-198): atomics sum in arrival order, a few ulps of the largest term
    ga, gb = native.chamfer_backward(_t(a), _t(b), _t(j1), _t(j2), _t(g1), _t(g2), deterministic=False)
    np.testing.assert_allclose(ga.cpu().numpy(), ra, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gb.cpu().numpy(), rb, rtol=1e-5, atol=1e-5)
    # the deterministic form sums in the order of the oracle's scalar loops: bit-exact, and the same on every run
    da, db = native.chamfer_backward(_t(a), _t(b), _t(j1), _t(j2), _t(g1), _t(g2), deterministic=True)
    rows1 = np.stack([np.bincount(j2[i], minlength=n) for i in range(B)])      # how many points chose each point
    rows2 = np.stack([np.bincount(j1[i], minlength=m) for i in range(B)])
    for got, want, rows in ((da.cpu().numpy(), ra, rows1), (db.cpu().numpy(), rb, rows2)):
        short = rows <= 8                                  # lists of <= 8 entries: the oracle's order, bit for bit
        assert np.array_equal(got[short].view(np.uint32), want[short].view(np.uint32))
        if (~short).any():                                  # longer lists: fixed-point sum, equal to fp32 rounding
            assert np.abs(got[~short] - want[~short]).max() <= 1e-5 * np.abs(want[~short]).max() + 1e-7
    for _ in range(3):
        ea, eb = native.chamfer_backward(_t(a), _t(b), _t(j1), _t(j2), _t(g1), _t(g2), deterministic=True)
        assert torch.equal(ea, da) and torch.equal(eb, db)


def test_chamfer_backward_deterministic_on_collapsed_clouds(native, ref):
    """Rows of every length: a prediction collapsed onto a few target points sends thousands of points to one
    neighbour (the long-row path: unordered list, fixed-point sum by a whole wave), mixed with isolated points and points
    nobody chose; a full-size call (B = 4, 16384 x 16384) for the LDS limits.  Points whose list has at most 8 entries
    must equal the oracle bit for bit (same order of additions); the others to fp32 rounding; every run the same bits."""
    rng = np.random.default_rng(99)
    for B, n, m, centres in ((2, 5000, 3000, 3), (1, 16384, 16384, 40), (4, 16384, 16384, 0), (2, 70, 16384, 1)):
        b = (rng.random((B, m, 3), dtype=np.float32) - 0.5)
        if centres:
            pick = rng.integers(0, m, (B, centres))
            a = np.stack([b[i][pick[i][rng.integers(0, centres, n)]] for i in range(B)]) + \
                rng.standard_normal((B, n, 3)).astype(np.float32) * 1e-3
            a[:, : n // 10] = rng.random((B, n // 10, 3), dtype=np.float32) - 0.5
        else:
            a = rng.random((B, n, 3), dtype=np.float32) - 0.5
        a = np.ascontiguousarray(a, dtype=np.float32)
        _, _, j1, j2 = ref.chamfer_forward(a, b)
        rows2 = np.stack([np.bincount(j1[i], minlength=m) for i in range(B)])      # list lengths of the cloud-2 points
        rows1 = np.stack([np.bincount(j2[i], minlength=n) for i in range(B)])
        if centres and n > 1000:
            assert rows2.max() > 64                                              # the long-row path is really taken
        g1 = rng.standard_normal((B, n)).astype(np.float32)
        g2 = rng.standard_normal((B, m)).astype(np.float32)
        ra, rb = ref.chamfer_backward(a, b, j1, j2, g1, g2)
        da, db = native.chamfer_backward(_t(a), _t(b), _t(j1), _t(j2), _t(g1), _t(g2), deterministic=True)
        ea, eb = native.chamfer_backward(_t(a), _t(b), _t(j1), _t(j2), _t(g1), _t(g2), deterministic=True)
        assert torch.equal(da, ea) and torch.equal(db, eb)
        da, db = da.cpu().numpy(), db.cpu().numpy()
        for got, want, rows in ((da, ra, rows1), (db, rb, rows2)):
            short = rows <= 8
            assert np.array_equal(got[short].view(np.uint32), want[short].view(np.uint32)), (B, n, m)
            if (~short).any():
                scale = np.abs(want[~short]).max()
                assert np.abs(got[~short] - want[~short]).max() <= 1e-5 * scale + 1e-7, (B, n, m)


FPS_CASES = [(2, 2048, 512, "uniform"), (3, 512, 256, "uniform"), (2, 2048, 256, "shell"), (2, 2048, 64, "shell"),
             (1, 100, 100, "uniform"), (2, 1500, 300, "dups"), (1, 4096, 128, "uniform"), (1, 16384, 64, "uniform"),
             (1, 700, 1, "uniform"), (2, 256, 16, "uniform")]


@pytest.mark.parametrize("B,N,npoint,kind", FPS_CASES)
def test_fps_exact(native, ref, B, N, npoint, kind):
    rng = np.random.default_rng(N + npoint)
    x = _cloud(rng, B, N, kind)
    got = native.furthest_point_sample(_t(x), npoint).cpu().numpy()
    want = ref.furthest_point_sample(x, npoint)
    assert np.array_equal(got, want)


def test_fps_all_zero_cloud(native, ref):
    x = np.zeros((2, 512, 3), np.float32)  # every candidate is skipped -> index 0 repeated
    got = native.furthest_point_sample(_t(x), 8).cpu().numpy()
    assert np.array_equal(got, ref.furthest_point_sample(x, 8))
    assert (got == 0).all()


def test_fps_prefix_consistent(native):
    x = torch.rand(2, 2048, 3, device="cuda") - 0.5
    full = native.furthest_point_sample(x, 512)
    assert torch.equal(full[:, :256], native.furthest_point_sample(x, 256))
    assert torch.equal(full[:, :64], native.furthest_point_sample(x, 64))


KNN_CASES = [(2, 2048, 2048, 16, "uniform"), (2, 2048, 512, 16, "uniform"), (2, 512, 512, 16, "uniform"),
             (2, 512, 256, 16, "uniform"), (2, 256, 256, 8, "uniform"), (2, 256, 576, 8, "uniform"),
             (2, 576, 64, 8, "uniform"), (2, 16384, 64, 32, "uniform"), (2, 2048, 2048, 16, "shell"),
             (2, 1000, 333, 16, "dups"), (1, 70, 5, 64, "uniform"), (1, 16, 3, 16, "uniform"),
             (1, 5000, 17, 1, "uniform"), (2, 3000, 100, 32, "dups")]


@pytest.mark.parametrize("B,N,S,k,kind", KNN_CASES)
def test_knn_exact(native, ref, B, N, S, k, kind):
    rng = np.random.default_rng(N * 7 + S * 3 + k)
    c, q = _cloud(rng, B, N, kind), _cloud(rng, B, S, kind)
    got = native.knn(k, _t(c), _t(q)).cpu().numpy()
    want = ref.knn(k, c, q)
    assert np.array_equal(got, want)


TOPK_CASES = [(16 * 512, 512, 30, "normal"), (2 * 512, 512, 64, "normal"), (64, 512, 512, "ties"), (300, 64, 64, "ties"),
              (77, 100, 7, "normal"), (50, 2048, 130, "ties"), (33, 1000, 65, "normal"), (9, 1, 1, "normal"),
              (40, 260, 1, "ties"), (21, 1023, 1023, "ties")]


@pytest.mark.parametrize("R,N,k,kind", TOPK_CASES)
def test_topk_rows_exact(native, ref, R, N, k, kind):
    """csrc/topk.hip vs the oracle: indices bit for bit, ties included ((value desc, column asc)); on tie-free rows
    that is torch.topk, which is what the reference calls (GDANet_util.py:5-11)."""
    rng = np.random.default_rng(R + 3 * N + k)
    x = rng.standard_normal((R, N)).astype(np.float32)
    if kind == "ties":
        x = (np.round(x * 2) / 2).astype(np.float32)
        x[0, : min(N, 4)] = [0.0, -0.0, np.inf, -np.inf][: min(N, 4)]
    got = native.topk_rows(_t(x), k)
    assert got.dtype == torch.int64 and tuple(got.shape) == (R, k)
    assert np.array_equal(got.cpu().numpy(), ref.topk_rows(x, k))
    if kind == "normal":
        assert torch.equal(got, _t(x).topk(k, dim=-1)[1])


def test_topk_rows_unaligned_rows_and_batch_shape(native, ref):
    rng = np.random.default_rng(11)
    flat = _t(rng.standard_normal(1 + 6 * 40 * 512).astype(np.float32))
    x = flat[1:].view(6, 40, 512)                         # 4-byte aligned only: the scalar-load variant
    assert x.data_ptr() % 16 != 0 and x.is_contiguous()
    got = native.topk_rows(x, 30)
    assert tuple(got.shape) == (6, 40, 30)
    assert np.array_equal(got.cpu().numpy(), ref.topk_rows(x.cpu().numpy(), 30))
    with pytest.raises(ValueError):
        native.topk_rows(x, 513)
    with pytest.raises(ValueError):
        native.topk_rows(x.transpose(1, 2), 3)


def test_gather_and_grad(native, ref):
    rng = np.random.default_rng(3)
    feat = rng.standard_normal((3, 35, 2048)).astype(np.float32)
    idx = ref.furthest_point_sample(_cloud(rng, 3, 2048), 512)
    out = native.gather_forward(_t(feat), _t(idx)).cpu().numpy()
    assert np.array_equal(out, ref.gather_forward(feat, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    g = native.gather_backward(_t(go), _t(idx), 2048).cpu().numpy()
    assert np.array_equal(g, ref.gather_backward(go, idx, 2048))  # FPS indices are unique: exact


@pytest.mark.parametrize("B,N,C,S,k", [(2, 16384, 3, 64, 32), (2, 256, 384, 576, 8), (1, 100, 7, 33, 5)])
def test_index_points(native, ref, B, N, C, S, k):
    rng = np.random.default_rng(N + C)
    pts = rng.standard_normal((B, N, C)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, S, k)).astype(np.int32)
    out = native.index_points_forward(_t(pts), _t(idx)).cpu().numpy()
    assert np.array_equal(out, ref.index_points_forward(pts, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    g = native.index_points_backward(_t(go), _t(idx), N).cpu().numpy()
    np.testing.assert_allclose(g, ref.index_points_backward(go, idx, N), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,C,Nk,Nq,k", [(2, 8, 2048, 2048, 16), (2, 32, 2048, 512, 16), (2, 64, 512, 512, 16),
                                         (2, 64, 512, 256, 16), (1, 5, 100, 37, 3), (1, 3, 50, 20, 8)])
def test_edge_group(native, ref, B, C, Nk, Nq, k):
    rng = np.random.default_rng(C + Nk + Nq)
    xk = rng.standard_normal((B, C, Nk)).astype(np.float32)
    xq = rng.standard_normal((B, C, Nq)).astype(np.float32)
    idx = rng.integers(0, Nk, size=(B, Nq, k)).astype(np.int32)
    out = native.edge_group_forward(_t(xk), _t(xq), _t(idx)).cpu().numpy()
    assert np.array_equal(out, ref.edge_group_forward(xk, xq, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    gk, gq = native.edge_group_backward(_t(go), _t(idx), Nk)
    rk, rq = ref.edge_group_backward(go, idx, Nk)
    np.testing.assert_allclose(gk.cpu().numpy(), rk, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(gq.cpu().numpy(), rq, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("B,N,C,S,k", [(2, 256, 384, 256, 8), (2, 576, 384, 576, 8), (1, 50, 6, 20, 4)])
def test_gather_max(native, ref, B, N, C, S, k):
    rng = np.random.default_rng(N + S)
    p = rng.standard_normal((B, N, C)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, S, k)).astype(np.int32)
    out, arg = native.gather_max_forward(_t(p), _t(idx))
    ro, ra = ref.gather_max_forward(p, idx)
    assert np.array_equal(out.cpu().numpy(), ro)
    assert np.array_equal(arg.cpu().numpy(), ra)
    go = rng.standard_normal(ro.shape).astype(np.float32)
    g = native.gather_max_backward(_t(go), _t(ra), N).cpu().numpy()
    np.testing.assert_allclose(g, ref.gather_max_backward(go, ra, N), rtol=1e-5, atol=1e-5)
    # fixed-point accumulation (csrc/gather.hip): the same bits on every run, a float64 sum to fp32 rounding, heavy
    # collisions (every query choosing the same few source points) included; NaN / infinity poison what they touch
    hot = np.minimum(ra, 3).astype(np.int32)
    want = np.zeros((B, N, C), np.float64)
    for b in range(B):
        for s_ in range(S):
            np.add.at(want[b], (hot[b, s_], np.arange(C)), go[b, s_].astype(np.float64))
    runs = [native.gather_max_backward(_t(go), _t(hot), N) for _ in range(3)]
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    err = np.abs(runs[0].cpu().numpy().astype(np.float64) - want)
    assert err.max() <= 2.0 ** -22 * np.abs(want).max() + 1e-30, err.max()
    bad = go.copy()
    bad[0, 1, 2], bad[B - 1, 0, 0] = np.nan, np.inf
    gb = native.gather_max_backward(_t(bad), _t(hot), N).cpu().numpy()
    assert np.isnan(gb[0, hot[0, 1, 2], 2]) and np.isinf(gb[B - 1, hot[B - 1, 0, 0], 0])
    ok = np.ones_like(gb, dtype=bool)
    ok[0, hot[0, 1, 2], 2] = ok[B - 1, hot[B - 1, 0, 0], 0] = False
    np.testing.assert_allclose(gb[ok], runs[0].cpu().numpy()[ok], rtol=1e-5, atol=1e-5)


def test_cpu_tensor_is_refused(native):
    with pytest.raises(Exception):
        native.chamfer_forward(torch.rand(1, 8, 3), torch.rand(1, 8, 3))


@pytest.mark.parametrize("B,n,eps,iters,kind", [(2, 1024, 0.005, 50, "uniform"), (1, 2048, 0.005, 100, "uniform"),
                                               (3, 1024, 0.002, 200, "dups"), (1, 4096, 0.005, 20, "uniform")])
def test_emd_forward_backward(native, ref, B, n, eps, iters, kind):
    from imgadapointr_amd.extensions.emd.emd_module import emdModule
    rng = np.random.default_rng(n + iters)
    a = _cloud(rng, B, n, kind) + 0.5     # the auction assumes coordinates in [0, 1]
    b = _cloud(rng, B, n, "uniform") + 0.5
    xa = _t(a).requires_grad_(True)
    dist, assignment = emdModule()(xa, _t(b), eps, iters)
    rd, ra = ref.emd_forward(a, b, eps, iters)
    assert np.array_equal(assignment.cpu().numpy(), ra)
    assert np.array_equal(dist.detach().cpu().numpy(), rd)
    w = rng.standard_normal((B, n)).astype(np.float32)
    (dist * _t(w)).sum().backward()
    np.testing.assert_allclose(xa.grad.cpu().numpy(), ref.emd_backward(a, b, w, ra), rtol=1e-6, atol=1e-7)


def test_emd_rejects_bad_shapes(native):
    from imgadapointr_amd.extensions.emd.emd_module import emdModule
    with pytest.raises(Exception):
        emdModule()(torch.rand(1, 1000, 3, device="cuda"), torch.rand(1, 1000, 3, device="cuda"), 0.005, 10)
    with pytest.raises(Exception):
        emdModule()(torch.rand(1, 1024, 3, device="cuda"), torch.rand(1, 2048, 3, device="cuda"), 0.005, 10)


def test_metrics_with_emd(native, ref):
    from imgadapointr_amd.utils.metrics import Metrics
    rng = np.random.default_rng(5)
    pred, gt = rng.random((1, 1024, 3), dtype=np.float32), rng.random((1, 1024, 3), dtype=np.float32)
    vals = [v.item() for v in Metrics.get(_t(pred), _t(gt), require_emd=True)]
    rd, _ = ref.emd_forward(pred, gt, 0.005, 100)
    assert abs(vals[3] - float(np.sqrt(rd).mean()) * 1000) < 1e-3
    assert abs(vals[1] - ref.chamfer_l1(pred, gt) * 1000) < 1e-3
    d1, d2, _, _ = ref.chamfer_forward(pred, gt)
    p, r = (np.sqrt(d1) < 0.01).mean(), (np.sqrt(d2) < 0.01).mean()
    assert abs(vals[0] - (2 * p * r / (p + r) if p + r else 0.0)) < 1e-6


@pytest.mark.parametrize("Nq,Nk,split,limit", [(576, 576, 512, 512), (576, 256, 0, 256), (256, 256, 0, 256),
                                                (512, 512, 0, 512), (96, 64, 32, 32), (32, 32, 0, 32),
                                                (160, 224, 64, 96), (64, 608, 32, 576)])
@pytest.mark.parametrize("bwd", ["fused", "split"])
def test_attention_forward_and_backward(native, Nq, Nk, split, limit, bwd, monkeypatch):
    """fp32 MFMA attention vs an explicit float64 softmax(q k^T scale + mask) v, values and gradients
    (both backward implementations: single-pass and the dq + dk/dv pair)."""
    from imgadapointr_amd.models import Transformer_utils as TU
    from imgadapointr_amd.models.Transformer_utils import _MFMAAttention
    monkeypatch.setattr(TU, "USE_MFMA_ATTENTION_BWD", bwd)
    torch.manual_seed(Nq + Nk)
    B, H, d = 3, 6, 64
    qkv = torch.randn(B, Nq, 3, H, d, device="cuda", requires_grad=True)     # strided q view, as in Attention
    kv = torch.randn(B, Nk, 2, H, d, device="cuda", requires_grad=True)
    q = qkv.permute(2, 0, 3, 1, 4)[0]
    k, v = kv.permute(2, 0, 3, 1, 4)[0], kv.permute(2, 0, 3, 1, 4)[1]
    scale = d ** -0.5
    assert native.attention_supported(q, k, split, limit)
    out = _MFMAAttention.apply(q, k, v, split, limit, scale)                  # [B,Nq,H,d]
    go = torch.randn_like(out)
    g = torch.autograd.grad(out, (qkv, kv), go)

    q64, k64, v64 = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    a = (q64 @ k64.transpose(-1, -2)) * scale
    if split:
        blocked = torch.zeros(Nq, Nk, dtype=torch.bool, device="cuda")
        blocked[:split, limit:] = True
        a = a.masked_fill(blocked, float("-inf"))
    gold = a.softmax(-1) @ v64                                                # [B,H,Nq,d]
    gq, gk, gv = torch.autograd.grad(gold, (q64, k64, v64), go.permute(0, 2, 1, 3).double())
    # fp32 accumulation over <= 576 keys: a few 1e-6 absolute on O(1) values
    assert (out.permute(0, 2, 1, 3).double() - gold).abs().max().item() < 5e-6
    assert (g[0][:, :, 0].double() - gq.permute(0, 2, 1, 3)).abs().max().item() < 2e-5
    assert (g[0][:, :, 1:].abs().max().item()) == 0.0                        # only the q slice receives gradient
    assert (g[1][:, :, 0].double() - gk.permute(0, 2, 1, 3)).abs().max().item() < 2e-5
    assert (g[1][:, :, 1].double() - gv.permute(0, 2, 1, 3)).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,Nq,Nk,split,limit", [(32, 576, 576, 512, 512), (32, 576, 256, 0, 256), (32, 256, 256, 0, 256),
                                                  (16, 256, 256, 0, 256), (32, 512, 512, 0, 512)])
def test_attention_backward_at_training_batch(native, B, Nq, Nk, split, limit, monkeypatch):
    """The production launch configurations (192 / 96 heads) of the single-pass kernel - automatic choice, query-range
    split forced off and forced on - against the independent dq + dk/dv kernel pair."""
    torch.manual_seed(B + Nq)
    H, d = 6, 64
    q = torch.randn(B, H, Nq, d, device="cuda")
    k = torch.randn(B, H, Nk, d, device="cuda")
    v = torch.randn(B, H, Nk, d, device="cuda")
    scale = d ** -0.5
    out, lse = native.attention_forward(q, k, v, split, limit, scale)
    go = torch.randn_like(out)
    res = []
    for fused, qsplit in ((False, None), (True, None), (True, "1"), (True, "2")):
        dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
        if qsplit is not None:
            monkeypatch.setenv("PTOPS_ATTN_QSPLIT", qsplit)
        native.attention_backward(q, k, v, out, go, lse, split, limit, scale, dq, dk, dv, fused=fused)
        monkeypatch.delenv("PTOPS_ATTN_QSPLIT", raising=False)
        res.append((dq, dk, dv))
    for got in res[1:]:
        for a, b in zip(got, res[0]):
            assert bool(torch.isfinite(a).all())
            assert float((a - b).abs().max()) < 3e-5 * max(1.0, float(b.abs().max()))
    # the split only changes where dk / dv are summed: dq is bit-identical, and forcing it on/off is deterministic
    assert torch.equal(res[2][0], res[3][0])


def test_attention_rejects_unsupported_shapes(native):
    q = torch.randn(1, 2, 40, 64, device="cuda")
    assert not native.attention_supported(q, q, 0, 40)                        # 40 is not a multiple of 32
    with pytest.raises(Exception):
        native.attention_forward(q, q, q, 0, 40, 0.125)
    q48 = torch.randn(1, 2, 64, 48, device="cuda")
    assert not native.attention_supported(q48, q48, 0, 64)


def test_packed_self_attention_and_linear_bias_grad(native):
    """Attention module on the packed-qkv path + Linear with the streaming bias-gradient kernel vs plain torch."""
    from imgadapointr_amd.models import Transformer_utils as TU
    torch.manual_seed(5)
    attn = TU.Attention(384, num_heads=6, qkv_bias=True).cuda()
    x = torch.randn(8, 576, 384, device="cuda", requires_grad=True)
    mask = torch.zeros(576, 576, device="cuda")
    mask[:512, 512:] = 1.
    outs = {}
    default_bwd = TU.USE_MFMA_ATTENTION_BWD
    for fast, native_bwd in ((True, "fused"), (True, "split"), (True, False), (False, False)):
        TU.USE_MFMA_ATTENTION = TU.FAST_BIAS_GRAD = fast
        TU.USE_MFMA_ATTENTION_BWD = native_bwd
        for p in attn.parameters():
            p.grad = None
        y = attn(x, mask=mask, mask_split=(512, 512))
        g = torch.autograd.grad(y.square().sum(), [x] + list(attn.parameters()))
        outs[(fast, native_bwd)] = [y.detach()] + [t.detach() for t in g]
    TU.USE_MFMA_ATTENTION = TU.FAST_BIAS_GRAD = True
    TU.USE_MFMA_ATTENTION_BWD = default_bwd
    for key in ((True, "fused"), (True, "split"), (True, False)):
        for a, b in zip(outs[key], outs[(False, False)]):
            scale = max(1.0, b.abs().max().item())
            assert (a - b).abs().max().item() < 2e-5 * scale, key
    # the column-sum kernel itself, against float64
    g2 = torch.randn(18432, 384, device="cuda")
    assert (native.colsum(g2).double() - g2.double().sum(0)).abs().max().item() < 2e-3
    g3 = torch.randn(2050, 96, device="cuda")
    assert (native.colsum(g3).double() - g3.double().sum(0)).abs().max().item() < 5e-4
    # repeated calls reuse one workspace (ticket words return to zero) and are bit-reproducible
    first = native.colsum(g2)
    for _ in range(5):
        assert torch.equal(native.colsum(g2), first)
    g4 = torch.randn(8192, 1536, device="cuda")
    assert (native.colsum(g4).double() - g4.double().sum(0)).abs().max().item() < 2e-3
    assert torch.equal(native.colsum(g3), native.colsum(g3))


@pytest.mark.parametrize("B,M,C", [(32, 576, 1024), (1, 512, 1024), (2, 256, 384), (3, 7, 4), (2, 1, 8), (1, 33, 100)])
def test_max_over_tokens_matches_torch(native, B, M, C):
    """libptops' max over tokens (values bit-equal to torch.max(dim=1), gradient to the arg-max token, lowest token on
    ties, NaN wins) — the memset-free replacement of ATen's strided reduction in the global-feature pools."""
    from imgadapointr_amd.models import Transformer_utils as TU
    torch.manual_seed(B * 100 + M + C)
    x = torch.randn(B, M, C, device="cuda")
    x[0, M // 2, 0] = 100.0
    if M > 2:
        x[:, 1, 1] = 50.0
        x[:, 2, 1] = 50.0                                        # exact tie: token 1 must take the gradient
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = TU.max_over_tokens(xa)
    assert type(ya.grad_fn).__name__.startswith("_MaxTokensFn")
    yb = torch.max(xb, dim=1)[0]
    assert torch.equal(ya, yb)
    g = torch.randn(B, C, device="cuda")
    ya.backward(g)
    assert torch.equal(xa.grad.sum(1), g) and int((xa.grad != 0).sum()) <= B * C
    _, arg = native.max_tokens_forward(x)
    assert torch.equal(torch.gather(x, 1, arg.long().unsqueeze(1)).squeeze(1), yb)
    if M > 2:
        assert bool((arg[:, 1] == 1).all())
        assert torch.equal(xa.grad[:, 1, 1], g[:, 1]) and float(xa.grad[:, 2, 1].abs().max()) == 0.0
    xn = x.clone()
    xn[0, M - 1, 0] = float("nan")
    out, _ = native.max_tokens_forward(xn)
    assert bool(torch.isnan(out[0, 0])) and torch.equal(out[0, 1:], yb[0, 1:])


@pytest.mark.parametrize("M,N,K", [(18432, 384, 384), (16384, 1152, 384), (8192, 768, 1024), (300, 96, 64), (129, 130, 16),
                                   (4097, 1027, 48)])
def test_gemm_split3_is_fp32_accurate(native, M, N, K):
    """fp32 GEMM on the bf16 matrix pipe (three bf16 pieces per operand, six exact piece products, fp32 accumulation):
    its error against float64 is of the order of the library's fp32 MFMA GEMM (both are bounded by the fp32 accumulation;
    asserted: within 2x of the library's error or 2e-6 of the output range for x.W^T and dy.W, within 4x or 1e-5 for
    dy^T.x - the claim made in bench.py / DESIGN 4.10a is exactly this), in all three operand layouts and on ragged tiles."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda") * torch.logspace(-3, 3, K, device="cuda")      # columns of very different size
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda")
    rows = slice(0, min(M, 1024))

    def rel(got, ref):
        return float((got.double() - ref).abs().max() / ref.abs().max())

    ref = x[rows].double() @ w.double().t() + b.double()
    got = native.gemm_split3(x, w, b)
    lib = torch.nn.functional.linear(x, w, b)
    assert rel(got[rows], ref) <= max(2.0 * rel(lib[rows], ref), 2e-6), (rel(got[rows], ref), rel(lib[rows], ref))
    ref = dy[rows].double() @ w.double()
    got = native.gemm_split3(dy, w, None, True, False) if N % 16 == 0 else None
    if got is not None:
        assert rel(got[rows], ref) <= max(2.0 * rel((dy @ w)[rows], ref), 2e-6)
    if M % 16 == 0:
        ref = dy.double().t() @ x.double()
        got = native.gemm_split3(dy, x, None, False, False)
        assert rel(got, ref) <= max(4.0 * rel(dy.t() @ x, ref), 1e-5)
    # row-sliced views (row stride != row length) and the alignment check
    xv = torch.randn(M, K + 16, device="cuda")[:, :K]
    assert rel(native.gemm_split3(xv, w)[rows], xv[rows].double() @ w.double().t()) < 5e-6
    with pytest.raises(native.PtopsError):
        native.gemm_split3(torch.randn(M, 20, device="cuda")[:, 1:17], w[:, :16])      # rows start 4 bytes off a 16-byte boundary


@pytest.mark.parametrize("M,N,K", [(18432, 384, 384), (8192, 1152, 384), (2048, 200, 132), (16, 128, 128), (4096, 4, 4),
                                   (65536, 512, 1024)])
def test_gemm_split3_wgrad_and_bias_gradient(native, M, N, K):
    """dW = dy^T.x with the contraction split over workgroups, and db = column sums of dy from the same pass: fp32
    accuracy against float64 (not worse than the library GEMM / torch.sum), ragged tiles, row-sliced views,
    bit-identical repeats (fixed reduction order)."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda") * torch.logspace(-2, 2, K, device="cuda")
    dy = torch.randn(M, N, device="cuda") + 0.25

    def rel(got, ref):
        return float((got.double() - ref).abs().max() / ref.abs().max())

    ref_w, ref_b = dy.double().t() @ x.double(), dy.double().sum(0)
    dw, db = native.gemm_split3_wgrad(dy, x, True)
    assert rel(dw, ref_w) <= max(4.0 * rel(dy.t() @ x, ref_w), 1e-5), (rel(dw, ref_w), rel(dy.t() @ x, ref_w))
    assert rel(db, ref_b) <= max(4.0 * rel(dy.sum(0), ref_b), 1e-5), (rel(db, ref_b), rel(dy.sum(0), ref_b))
    dw2, db2 = native.gemm_split3_wgrad(dy, x, True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    dw3, none = native.gemm_split3_wgrad(dy, x, False)
    assert none is None and torch.equal(dw, dw3)
    xv = torch.randn(M, K + 8, device="cuda")[:, 4:4 + K]
    dyv = torch.randn(M, N + 4, device="cuda")[:, :N]
    dwv, dbv = native.gemm_split3_wgrad(dyv, xv, True)
    assert rel(dwv, dyv.double().t() @ xv.double()) < 2e-5 and rel(dbv, dyv.double().sum(0)) < 2e-5
    if M > 16:
        with pytest.raises(native.PtopsError):
            native.gemm_split3_wgrad(dy[:M - 8], x[:M - 8])             # M % 16 != 0


@pytest.mark.parametrize("B,N,S,C,k", [(32, 640, 640, 384, 8), (3, 256, 100, 384, 10), (2, 77, 50, 6, 5)])
def test_gather_max_act_equals_the_three_ops(native, B, N, S, C, k):
    """LeakyReLU(max_j p[idx] + add) in one launch each way (`ptops_gather_max_act_*`) against gather_max, `+` and
    F.leaky_relu as separate ops: outputs and all gradients bit for bit (the same float operations; exact zeros of the
    pre-activation take the slope branch in both)."""
    from imgadapointr_amd.models import Transformer_utils as TU
    torch.manual_seed(B + N + C)
    p0 = torch.randn(B, N, C, device="cuda")
    add0 = torch.randn(B, S, C, device="cuda")
    idx = torch.randint(0, N, (B, S, k), device="cuda", dtype=torch.int32)
    add0[0, 0, :] = -p0[0].index_select(0, idx[0, 0].long()).max(0).values            # pre-activation exactly 0
    w = torch.randn(B, S, C, device="cuda")
    res = {}
    for fused in (True, False):
        TU.FUSED_GATHER_MAX_ACT = fused
        try:
            p, add = p0.clone().requires_grad_(True), add0.clone().requires_grad_(True)
            out = TU.gather_max_add_leaky_relu(p, idx, add, 0.2)
            (out * w).sum().backward()
            res[fused] = (out.detach(), p.grad, add.grad)
        finally:
            TU.FUSED_GATHER_MAX_ACT = True
    assert float(res[True][0][0, 0].abs().max()) == 0.0
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    with pytest.raises(native.PtopsError):
        native.gather_max_act_backward(torch.zeros(1, 4, 4, device="cuda"), torch.zeros(1, 4, 4, device="cuda"),
                                       torch.zeros(1, 4, 4, device="cuda", dtype=torch.int32), 0.2, 2000)


def test_grouped_weight_gradients(native):
    """`ptops_gemm_f32_split3_wgrad_grouped`: the dW / db of a ragged mix of layers (tiles that do not fill 128 x 128,
    contractions that do not divide into equal jobs, row-sliced operands, with and without db, a one-job problem) from ONE
    grouped launch against float64 - not worse than the library GEMM / torch.sum - bit-identical across runs, for
    several job lengths, and equal to the per-layer kernel's accuracy class; the C side refuses a plan that does not
    belong to the problems."""
    import ctypes
    import numpy as np
    torch.manual_seed(77)
    shapes = [(18432, 384, 384, True), (8192, 1152, 384, False), (2048, 200, 132, True), (16, 128, 128, True),
              (4096, 4, 4, True), (4624, 96, 512, True), (8192, 384, 768, False), (18432, 384, 384, True)]
    probs = []
    for M, N, K, wb in shapes:
        x = (torch.randn(M, K + 8, device="cuda") * torch.logspace(-2, 2, K + 8, device="cuda"))[:, 4:4 + K]
        dy = (torch.randn(M, N + 4, device="cuda") + 0.25)[:, :N]
        probs.append((dy, x, wb))

    def rel(got, ref):
        return float((got.double() - ref).abs().max() / ref.abs().max())

    def run():
        outs = [native.wgrad_defer(dy, x, wb) for dy, x, wb in probs]
        native.wgrad_flush()
        return outs

    first = None
    for target in (0, 24, 100, 4096):
        native.set_option("PTOPS_WGRAD_GROUP_STEPS", target)
        native._wgrad_plans.clear()
        try:
            outs = run()
            again = run()
        finally:
            native.set_option("PTOPS_WGRAD_GROUP_STEPS", 0)
            native._wgrad_plans.clear()
        for (dy, x, wb), (dw, db), (dw2, db2) in zip(probs, outs, again):
            ref_w = dy.double().t() @ x.double()
            assert rel(dw, ref_w) <= max(4.0 * rel(dy.t() @ x, ref_w), 1e-5), (target, tuple(dy.shape), rel(dw, ref_w))
            assert torch.equal(dw, dw2)
            assert (db is not None) == wb
            if wb:
                ref_b = dy.double().sum(0)
                assert rel(db, ref_b) <= max(4.0 * rel(dy.sum(0), ref_b), 1e-5), (target, tuple(dy.shape), rel(db, ref_b))
                assert torch.equal(db, db2)
        first = first or outs
    # the same layer twice in one group: two independent, equal results
    assert torch.equal(first[0][0], first[7][0]) is False                  # different random operands ...
    dup = [probs[0], probs[0]]
    outs = [native.wgrad_defer(dy, x, wb) for dy, x, wb in dup]
    native.wgrad_flush()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # plan / problem mismatch is refused on the host, before anything is launched
    lib = native.load()
    mnk = np.array([[4096, 128, 128]], dtype=np.int32)
    need = lib.ptops_wgrad_group_plan(mnk.ctypes.data, None, 1, 0, None, 0)
    plan = np.zeros(need, dtype=np.uint8)
    assert lib.ptops_wgrad_group_plan(mnk.ctypes.data, None, 1, 0, plan.ctypes.data, need) == need
    dy, x = torch.randn(2048, 128, device="cuda"), torch.randn(2048, 128, device="cuda")
    dw = torch.empty(128, 128, device="cuda")
    row = np.array([[dy.data_ptr(), x.data_ptr(), dw.data_ptr(), 0, 128, 128, 128, 2048 | (128 << 32), 128]], dtype=np.int64)
    ws = torch.empty(int(lib.ptops_wgrad_group_workspace_bytes(plan.ctypes.data)), dtype=torch.uint8, device="cuda")
    rc = lib.ptops_gemm_f32_split3_wgrad_grouped(row.ctypes.data, torch.from_numpy(row).cuda().data_ptr(), 1, plan.ctypes.data,
                                                 torch.from_numpy(plan).cuda().data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc != 0 and b"plan" in lib.ptops_last_error()
    assert lib.ptops_wgrad_group_plan(np.array([[100, 8, 8]], dtype=np.int32).ctypes.data, None, 1, 0, None, 0) == -1   # M % 16


def test_linear_layers_on_split3_match_library(native):
    """nn.Linear-compatible module: forward, input gradient, weight gradient and bias gradient with the split-bf16 GEMM
    switched on and off agree to fp32 accuracy (each is compared with float64)."""
    from imgadapointr_amd.models import Transformer_utils as TU
    torch.manual_seed(21)
    lin = TU.Linear(384, 1152).cuda()
    x = torch.randn(32, 576, 384, device="cuda", requires_grad=True)
    g = torch.randn(32, 576, 1152, device="cuda")
    res = {}
    for on in (True, False):
        TU.USE_SPLIT3_GEMM = on
        lin.zero_grad()
        y = lin(x)
        gx, = torch.autograd.grad(y, x, g, retain_graph=True)
        y.backward(g)
        res[on] = (y.detach(), gx, lin.weight.grad.clone(), lin.bias.grad.clone())
    TU.USE_SPLIT3_GEMM = True
    ref_y = x.detach().double() @ lin.weight.detach().double().t() + lin.bias.detach().double()
    ref_gx = g.double() @ lin.weight.detach().double()
    for a, bq, ref in ((res[True][0], res[False][0], ref_y), (res[True][1], res[False][1], ref_gx)):
        ea, eb = float((a.double() - ref).abs().max()), float((bq.double() - ref).abs().max())
        assert ea <= max(2.0 * eb, 1e-6 * float(ref.abs().max())), (ea, eb)
    g2, x2 = g.view(-1, 1152).double(), x.detach().view(-1, 384).double()
    for a, bq, ref in ((res[True][2], res[False][2], g2.t() @ x2), (res[True][3], res[False][3], g2.sum(0))):
        ea, eb = float((a.double() - ref).abs().max()), float((bq.double() - ref).abs().max())
        assert ea <= max(2.0 * eb, 1e-6 * float(ref.abs().max())), (ea, eb)


def test_bias_gradient_handoff_ignores_other_layouts(native):
    """Column sums a producer kernel offers for a gradient tensor are taken only for that very layout.
    A transposed view of a square gradient has the same address and shape but other strides; an in-place write after
    the offer changes the data: both must fall back to the column-sum kernel."""
    from imgadapointr_amd.models import Transformer_utils as TU
    torch.manual_seed(9)
    M = C = 2304                                           # square, >= 2048 rows: the _LinearFn path
    lin = TU.Linear(C, C).cuda()
    x = torch.randn(M, C, device="cuda", requires_grad=True)
    g = torch.randn(M, C, device="cuda")
    wrong = torch.full((C,), 123.0, device="cuda")         # what a stale hand-off would deliver
    want = g.t().double().sum(0)

    def bias_grad(grad_out):
        lin.bias.grad = None
        y = lin(x)
        assert type(y.grad_fn).__name__.startswith("_LinearFn")
        y.backward(grad_out)
        return lin.bias.grad.double()

    TU._clear_column_sums()
    TU._offer_column_sums(g, wrong)
    got = bias_grad(g.t())                                 # same storage, same shape, transposed strides
    assert (got - want).abs().max().item() < 2e-3, "took column sums that belong to another layout"
    TU._clear_column_sums()
    TU._offer_column_sums(g, wrong)
    g.mul_(1.0)                                            # version bump after the offer
    got = bias_grad(g)
    assert (got - g.double().sum(0)).abs().max().item() < 2e-3, "took column sums of a tensor written to since"
    TU._clear_column_sums()
    sums = native.colsum(g)
    TU._offer_column_sums(g, sums)
    assert torch.equal(bias_grad(g), sums.double())        # the intended hit still works
    assert not TU._column_sums                             # and is consumed


@pytest.mark.parametrize("B,N", [(3, 512), (2, 2048)])
def test_fused_edge_conv_matches_literal_chain(native, B, N):
    """DGCNN grouper: fused gather/statistics/select kernels vs the reference's literal
    gather+concat -> Conv2d -> GroupNorm -> LeakyReLU -> max chain; outputs and all gradients."""
    from imgadapointr_amd.models import base_blocks as BB
    from golden_util import seeded_state_dict, synthetic_batch
    g = BB.DGCNN_Grouper(k=16)
    sd = seeded_state_dict(g.state_dict(), "grouper-fused", std=0.3)
    for key in sd:                         # mixed-sign GroupNorm scales exercise the min branch
        if key.endswith("1.weight"):
            sd[key] = sd[key] * torch.where(torch.arange(sd[key].numel()) % 3 == 0, -1.0, 1.0)
    g.load_state_dict(sd)
    g = g.cuda()
    x = torch.from_numpy(synthetic_batch(B, N, 64, seed=N, structured=(N == 512))[0]).cuda()
    res = {}
    for fused in (True, False):
        BB.FUSED_EDGE_CONV = fused
        xin = x.clone().requires_grad_(True)
        coor, f = g(xin, [N // 4, N // 8])
        w = torch.linspace(-1, 1, f.numel(), device="cuda").view_as(f)
        grads = torch.autograd.grad((f * w).sum(), [xin] + list(g.parameters()))
        res[fused] = (coor.detach(), f.detach(), [t.detach() for t in grads])
    BB.FUSED_EDGE_CONV = True
    assert torch.equal(res[True][0], res[False][0])

    # the referee: the literal chain of the same grouper in float64 (kNN / FPS picks from the fp32 kernels, so all three
    # evaluations see the same graph), as in benchmarks/edgeconv_referee.py
    from imgadapointr_amd.models.Transformer_utils import knn_idx

    class G64(BB.DGCNN_Grouper):
        def edge_conv(self, layer, coor_q, x_q, coor_k, x_k):
            idx = knn_idx(self.k, coor_k.float().transpose(-1, -2).contiguous(),
                          coor_q.float().transpose(-1, -2).contiguous()).long()
            Bq, C, Nk = x_k.shape
            Nq = x_q.shape[2]
            nb = torch.gather(x_k.unsqueeze(2).expand(Bq, C, Nq, Nk), 3, idx.unsqueeze(1).expand(Bq, C, Nq, self.k))
            feat = torch.cat([nb - x_q.unsqueeze(-1), x_q.unsqueeze(-1).expand_as(nb)], dim=1)
            return layer(feat).max(dim=-1)[0]

        @staticmethod
        def fps_downsample(coor, x, num_group, fps_idx=None):
            idx = native.furthest_point_sample(coor.float().transpose(1, 2).contiguous(), num_group).long()
            comb = torch.cat([coor, x], dim=1)
            out = torch.gather(comb, 2, idx.unsqueeze(1).expand(-1, comb.size(1), -1))
            return out[:, :3], out[:, 3:]

    g64 = G64(k=16)
    g64.load_state_dict(sd)
    g64 = g64.cuda().double()
    xin = x.double().clone().requires_grad_(True)
    _, f64 = g64(xin, [N // 4, N // 8])
    w64 = torch.linspace(-1, 1, f64.numel(), device="cuda").view_as(f64).double()
    ref = [t.detach() for t in torch.autograd.grad((f64 * w64).sum(), [xin] + list(g64.parameters()))]

    fa, fb = res[True][1].double(), res[False][1].double()
    scale = max(1.0, f64.abs().max().item())
    assert (fa - f64).abs().max().item() < 2e-5 * scale and (fb - f64).abs().max().item() < 2e-5 * scale
    for (name, _), a, b, r in zip([("x", None)] + list(g.named_parameters()), res[True][2], res[False][2], ref):
        # A near-tie between two neighbours (top-2 gap below fp32 rounding) may select the other edge in ANY fp32
        # evaluation, fused or literal; that moves one O(1) contribution of this loss to another source point and fans
        # out through the earlier stages.  Each form is therefore judged against float64 (relative L2, flip-tolerant):
        # the fused kernels must be as close to it as the reference's literal chain is.
        # (a flip is a genuine O(1) change that fans out through the earlier stages, in whichever fp32 form it happens:
        # on one box the literal chain matched all of float64's selections (5.5e-7) and the fused one flipped one
        # (2.8e-3, 3 % of the elements of dx beyond 1e-4 of its range), so the bound is the flip-tolerant relative L2;
        # the element-wise check on identical inputs is test_fused_edge_conv_function_vs_float64)
        for form, t_ in (("fused", a), ("literal", b)):
            rel = ((t_.double() - r).norm() / r.norm().clamp_min(1e-12)).item()
            assert rel < 2e-2, (name, form, rel)


@pytest.mark.parametrize("B,C,Nq,Nk,k,G", [(2, 32, 300, 700, 16, 4), (3, 64, 512, 512, 8, 4), (1, 128, 256, 512, 16, 4)])
def test_fused_edge_conv_function_vs_float64(native, B, C, Nq, Nk, k, G):
    """_FusedEdgeConv on leaf P / R against the same function written with torch ops in float64."""
    from imgadapointr_amd.models.base_blocks import _FusedEdgeConv
    gen = torch.Generator().manual_seed(C + Nq)
    P = torch.randn(B, C, Nk, generator=gen).cuda().requires_grad_(True)
    R = torch.randn(B, C, Nq, generator=gen).cuda().requires_grad_(True)
    idx = torch.randint(0, Nk, (B, Nq, k), generator=gen, dtype=torch.int32).cuda()
    gamma = (torch.randn(C, generator=gen) * 0.7).cuda().requires_grad_(True)     # both signs
    beta = (torch.randn(C, generator=gen) * 0.3).cuda().requires_grad_(True)
    # coherent + random upstream gradient: the group-mean terms of the GroupNorm backward are not negligible
    w = (torch.randn(B, C, Nq, generator=gen) * 0.5).cuda() + \
        torch.linspace(-1, 1, B * C * Nq, device="cuda").view(B, Nq, C).transpose(1, 2)
    out = _FusedEdgeConv.apply(P, R, idx, gamma, beta, G, 1e-5, 0.2)
    got = torch.autograd.grad((out * w).sum(), [P, R, gamma, beta])

    Pd, Rd, gd, bd = [t.detach().double().requires_grad_(True) for t in (P, R, gamma, beta)]
    nb = torch.gather(Pd.unsqueeze(2).expand(B, C, Nq, Nk), 3, idx.long().unsqueeze(1).expand(B, C, Nq, k))
    y = torch.nn.functional.group_norm(nb + Rd.unsqueeze(-1), G, gd, bd, 1e-5)
    ref = torch.nn.functional.leaky_relu(y, 0.2).max(dim=-1)[0]
    want = torch.autograd.grad((ref * w.double()).sum(), [Pd, Rd, gd, bd])
    assert (out.double() - ref).abs().max().item() < 2e-5
    for name, a, b in zip(("dP", "dR", "dgamma", "dbeta"), got, want):
        tol = 1e-4 * max(1.0, b.abs().max().item())
        bad = ((a.double() - b).abs() > tol).float().mean().item()
        assert bad <= 1e-4, (name, bad, (a.double() - b).abs().max().item())


@pytest.mark.parametrize("rows,C", [(18432, 384), (8192, 384), (1000, 96), (515, 1024), (3, 8), (4096, 640)])
def test_layer_norm_forward_and_backward(native, rows, C):
    """Streaming LayerNorm kernels vs float64 torch: y, dx, dgamma, dbeta."""
    torch.manual_seed(rows + C)
    x = (torch.randn(rows, C, device="cuda") * 2 + 0.7).requires_grad_(True)
    w = (torch.randn(C, device="cuda") * 0.5 + 1).requires_grad_(True)
    b = torch.randn(C, device="cuda").requires_grad_(True)
    go = torch.randn(rows, C, device="cuda") + 0.1
    y, mean, rstd = native.layer_norm_forward(x.detach(), w.detach(), b.detach(), 1e-5)
    dx, dgb = native.layer_norm_backward(go, x.detach(), mean, rstd, w.detach())
    extra = torch.randn(rows, C, device="cuda")
    dx_e, dgb_e = native.layer_norm_backward(go, x.detach(), mean, rstd, w.detach(), extra)
    assert torch.equal(dgb_e[:2 * C], dgb[:2 * C])
    assert (dx_e - (dx + extra)).abs().max().item() <= 1e-6 * max(1.0, dx.abs().max().item())
    assert (dgb_e[2 * C:].double() - dx_e.double().sum(0)).abs().max().item() < 2e-3 * max(1.0, dx_e.abs().max().item())
    s_, y_, mean_, rstd_ = native.add_layer_norm_forward(x.detach(), extra, w.detach(), b.detach(), 1e-5)
    assert torch.equal(s_, x.detach() + extra)
    y2_, mean2_, rstd2_ = native.layer_norm_forward(s_, w.detach(), b.detach(), 1e-5)
    assert torch.equal(y_, y2_) and torch.equal(mean_, mean2_) and torch.equal(rstd_, rstd2_)
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = torch.nn.functional.layer_norm(x64, (C,), w64, b64, 1e-5)
    gx, gw, gb = torch.autograd.grad(ref, (x64, w64, b64), go.double())
    assert (y.double() - ref).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item())
    assert (dx.double() - gx).abs().max().item() < 1e-5 * max(1.0, gx.abs().max().item())
    assert (dgb[:C].double() - gw).abs().max().item() < 2e-5 * max(1.0, gw.abs().max().item())
    assert (dgb[C:2 * C].double() - gb).abs().max().item() < 2e-5 * max(1.0, gb.abs().max().item())
    assert (dgb[2 * C:].double() - dx.double().sum(0)).abs().max().item() < 2e-3 * max(1.0, dx.abs().max().item())
    # deterministic
    dx2, dgb2 = native.layer_norm_backward(go, x.detach(), mean, rstd, w.detach())
    assert torch.equal(dx, dx2) and torch.equal(dgb, dgb2)


def test_layer_norm_module_matches_torch(native):
    from imgadapointr_amd.models import Transformer_utils as TU
    torch.manual_seed(3)
    ln = TU.LayerNorm(384, eps=1e-6).cuda()
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.2)
        ln.bias.normal_(0.0, 0.2)
    x = torch.randn(32, 576, 384, device="cuda", requires_grad=True)
    res = {}
    for fast in (True, False):
        TU.FAST_LAYER_NORM = fast
        y = ln(x.transpose(0, 1).transpose(0, 1))          # a non-trivial view of x
        g = torch.autograd.grad(y.square().sum(), [x, ln.weight, ln.bias])
        res[fast] = [y.detach()] + [t.detach() for t in g]
    TU.FAST_LAYER_NORM = True
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() < 2e-5 * max(1.0, b.abs().max().item())
    assert sorted(ln.state_dict().keys()) == ["bias", "weight"]


def test_fused_add_layer_norm_in_blocks(native):
    """Decoder / encoder blocks with the residual add folded into the following LayerNorm vs the plain form."""
    from imgadapointr_amd.models import Transformer_utils as TU
    from imgadapointr_amd.models import base_blocks as BB
    torch.manual_seed(11)
    enc = BB.SelfAttnBlockApi(384, 6, mlp_ratio=2., qkv_bias=True, block_style='attn-graph', combine_style='concat', k=8).cuda()
    dec = BB.CrossAttnBlockApi(384, 6, mlp_ratio=2., qkv_bias=True, self_attn_block_style='attn-graph',
                               self_attn_combine_style='concat', cross_attn_block_style='attn-graph',
                               cross_attn_combine_style='concat', k=8).cuda()
    B, N, M = 8, 256, 128
    x = torch.randn(B, N, 384, device="cuda", requires_grad=True)
    v = torch.randn(B, M, 384, device="cuda", requires_grad=True)
    pos, vpos = torch.rand(B, N, 3, device="cuda"), torch.rand(B, M, 3, device="cuda")
    idx_s = TU.knn_point(8, pos, pos)
    idx_c = TU.knn_point(8, vpos, pos)
    res = {}
    params = list(enc.parameters()) + list(dec.parameters())
    for fused in (True, False):
        TU.FUSED_ADD_LAYER_NORM = fused
        h = enc(x, pos, idx=idx_s)
        y = dec(h, v, pos, vpos, self_attn_idx=idx_s, cross_attn_idx=idx_c)
        g = torch.autograd.grad(y.square().sum(), [x, v] + params)
        res[fused] = [y.detach()] + [t.detach() for t in g]
    TU.FUSED_ADD_LAYER_NORM = True
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() < 3e-5 * max(1.0, b.abs().max().item())



def test_fused_add_layer_norm_across_blocks(native):
    """TransformerEncoder / TransformerDecoder stacks: the last residual add of a block folded into the next block's
    norm1 (and every in-block fusion) vs the plain form; outputs and all gradients."""
    from imgadapointr_amd.models import Transformer_utils as TU
    from imgadapointr_amd.models import base_blocks as BB
    torch.manual_seed(12)
    enc = BB.TransformerEncoder(embed_dim=384, depth=3, num_heads=6, mlp_ratio=2., qkv_bias=True,
                                block_style_list=['attn-graph', 'attn', 'attn'], combine_style='concat', k=8).cuda()
    dec = BB.TransformerDecoder(embed_dim=384, depth=3, num_heads=6, mlp_ratio=2., qkv_bias=True,
                                self_attn_block_style_list=['attn-graph', 'attn', 'attn'], self_attn_combine_style='concat',
                                cross_attn_block_style_list=['attn-graph', 'attn', 'attn'], cross_attn_combine_style='concat',
                                k=8).cuda()
    B, N, M = 8, 256, 128
    x = torch.randn(B, M, 384, device="cuda", requires_grad=True)
    q = torch.randn(B, N, 384, device="cuda", requires_grad=True)
    pos, qpos = torch.rand(B, M, 3, device="cuda"), torch.rand(B, N, 3, device="cuda")
    params = list(enc.parameters()) + list(dec.parameters())
    res = {}
    for fused in (True, False):
        TU.FUSED_ADD_LAYER_NORM = fused
        mem = enc(x, pos)
        y = dec(q, mem, qpos, pos, denoise_length=64)
        g = torch.autograd.grad(y.square().sum(), [x, q] + params)
        res[fused] = [y.detach()] + [t.detach() for t in g]
    TU.FUSED_ADD_LAYER_NORM = True
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() < 5e-5 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("M,N,slope", [(18432, 1024, 0.2), (1000, 64, 0.2), (4099, 128, 1.0)])
def test_batch_norm_rows_matches_torch(native, M, N, slope):
    """Column-moment BatchNorm (+ LeakyReLU) kernels vs nn.BatchNorm1d in float64, incl. running statistics."""
    from imgadapointr_amd.models.AdaPoinTr import _BatchNormRowsLeakyReLU
    torch.manual_seed(M + N)
    x = (torch.randn(M, N, device="cuda") * 1.7 + 0.4).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(N).cuda().train()
    with torch.no_grad():
        bn.weight.normal_(1.0, 0.3)
        bn.bias.normal_(0.0, 0.3)
    ref_bn = torch.nn.BatchNorm1d(N).cuda().double().train()
    ref_bn.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    go = torch.randn(M, N, device="cuda")
    y = _BatchNormRowsLeakyReLU.apply(x, bn.weight, bn.bias, bn, slope)
    gx, gw, gb = torch.autograd.grad(y, (x, bn.weight, bn.bias), go)
    x64 = x.detach().double().requires_grad_(True)
    r = torch.nn.functional.leaky_relu(ref_bn(x64), slope)
    rx, rw, rb = torch.autograd.grad(r, (x64, ref_bn.weight, ref_bn.bias), go.double())
    assert (y.double() - r).abs().max().item() < 2e-5
    assert (gx.double() - rx).abs().max().item() < 2e-5 * max(1.0, rx.abs().max().item())
    assert (gw.double() - rw).abs().max().item() < 1e-4 * max(1.0, rw.abs().max().item())
    assert (gb.double() - rb).abs().max().item() < 1e-4 * max(1.0, rb.abs().max().item())
    assert (bn.running_mean.double() - ref_bn.running_mean).abs().max().item() < 1e-6
    assert (bn.running_var.double() - ref_bn.running_var).abs().max().item() < 1e-5
    assert int(bn.num_batches_tracked) == 1


def test_gelu_backward_with_column_sums(native):
    """Fused GELU backward (+ fc1 bias gradient) vs plain torch in an Mlp: outputs and all gradients."""
    from imgadapointr_amd.models import Transformer_utils as TU
    torch.manual_seed(21)
    mlp = TU.Mlp(384, 768).cuda()
    x = torch.randn(32, 576, 384, device="cuda", requires_grad=True)
    res = {}
    for fused in (True, False):
        TU.FUSED_GELU_BACKWARD = fused
        y = mlp(x)
        g = torch.autograd.grad(y.square().sum(), [x] + list(mlp.parameters()))
        res[fused] = [y.detach()] + [t.detach() for t in g]
    TU.FUSED_GELU_BACKWARD = True
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() < 2e-5 * max(1.0, b.abs().max().item())
    xx = torch.randn(3000, 96, device="cuda") * 3
    gg = torch.randn(3000, 96, device="cuda")
    dpre, cs = native.gelu_backward_colsum(xx, gg)
    x64 = xx.double().requires_grad_(True)
    (want,) = torch.autograd.grad(torch.nn.functional.gelu(x64), x64, gg.double())
    assert (dpre.double() - want).abs().max().item() < 2e-6
    assert (cs.double() - want.sum(0)).abs().max().item() < 1e-3


@pytest.mark.parametrize("self_attention", [True, False])
def test_padded_head_multihead_attention_equals_module(native, self_attention):
    """nn.MultiheadAttention(384, 8) (48-wide heads) through the 64-wide MFMA attention with zero-padded projection
    weights vs the module's own forward, in float64: output and every gradient."""
    from imgadapointr_amd.models import pctransformer as PT
    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    B, N, M, E = 8, 256, 256 if self_attention else 512, 384
    mha = torch.nn.MultiheadAttention(E, 8).to(dev)
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.1)
        mha.out_proj.bias.normal_(0, 0.1)
    x = torch.randn(B, N, E, device=dev, requires_grad=True)
    ctx = x if self_attention else torch.randn(B, M, E, device=dev, requires_grad=True)
    up = torch.randn(B, N, E, device=dev)

    y = PT.multihead_attention(mha, x, ctx)
    y.backward(up)
    got = [y.detach()] + [p.grad.clone() for p in (x, *((ctx,) if not self_attention else ()), *mha.parameters())]

    ref = torch.nn.MultiheadAttention(E, 8).to(dev).double()
    ref.load_state_dict({k: v.double() for k, v in mha.state_dict().items()})
    xd = x.detach().double().requires_grad_(True)
    cd = xd if self_attention else ctx.detach().double().requires_grad_(True)
    yd = ref(xd.transpose(0, 1), cd.transpose(0, 1), cd.transpose(0, 1), need_weights=False)[0].transpose(0, 1)
    yd.backward(up.double())
    want = [yd.detach()] + [p.grad for p in (xd, *((cd,) if not self_attention else ()), *ref.parameters())]
    for g, w in zip(got, want):
        err = float((g.double() - w).abs().max() / w.abs().max())
        assert err < 2e-5, err
    # the switch really selects the other path
    PT.PADDED_HEAD_MFMA_ATTENTION = False
    try:
        y2 = PT.multihead_attention(mha, x, ctx)
    finally:
        PT.PADDED_HEAD_MFMA_ATTENTION = True
    assert float((y2 - y).detach().abs().max()) < 1e-4 and not torch.equal(y2, y)


def test_full_size_properties(native):
    """BASELINE sizes (B = 32, 2048 -> 16384), where the CPU oracle would take minutes: properties instead.
    Chamfer: three independent algorithms (brute force, pruned sweep, MFMA filter) agree bit for bit, the two directions are each
    other's mirror, distances are invariant under a permutation of the target and the indices follow it, and the
    returned index reproduces the returned distance.  FPS: prefix property and first index.  kNN: the k-th distance
    equals torch.topk's on the reference's own expansion-form matrix up to its rounding, neighbours are sorted."""
    torch.manual_seed(7)
    B, n, m = 32, 16384, 16384
    a = torch.rand(B, n, 3, device="cuda") - 0.5
    b = a[:, torch.randperm(n, device="cuda")] + torch.randn(B, m, 3, device="cuda") * 0.01
    brute = native.chamfer_forward(a, b, algorithm=native.CHAMFER_BRUTE)
    sweep = native.chamfer_forward(a, b, algorithm=native.CHAMFER_SWEEP)
    assert all(torch.equal(x, y) for x, y in zip(brute, sweep))
    mfma = native.chamfer_forward(a, b, algorithm=native.CHAMFER_MFMA)
    assert all(torch.equal(x, y) for x, y in zip(brute, mfma))
    d1, d2, i1, i2 = brute
    m1, m2, j1, j2 = native.chamfer_forward(b, a, algorithm=native.CHAMFER_BRUTE)
    assert torch.equal(d1, m2) and torch.equal(d2, m1) and torch.equal(i1, j2) and torch.equal(i2, j1)
    # the index reproduces the distance (same subtraction order as the kernel: query - target)
    near = torch.gather(b, 1, i1.long().unsqueeze(-1).expand(-1, -1, 3))
    diff = a - near
    recomputed = diff[..., 0] * diff[..., 0] + (diff[..., 1] * diff[..., 1] + diff[..., 2] * diff[..., 2])
    assert float((recomputed - d1).abs().max()) <= 1e-9 + 4 * float(d1.max()) * 2 ** -23
    perm = torch.randperm(m, device="cuda")
    p1, p2, pi1, pi2 = native.chamfer_forward(a, b[:, perm], algorithm=native.CHAMFER_SWEEP)
    assert torch.equal(p1, d1) and torch.equal(p2, d2[:, perm])
    same = perm[pi1.long()] == i1.long()                       # an exact distance tie may resolve to another index
    assert float(same.float().mean()) > 0.9999

    part = torch.rand(B, 2048, 3, device="cuda") - 0.5
    f512 = native.furthest_point_sample(part, 512)
    f64 = native.furthest_point_sample(part, 64)
    assert torch.equal(f512[:, :64], f64) and bool((f512[:, 0] == 0).all())
    assert all(len(set(row.tolist())) == 512 for row in f512[:4].cpu())

    k = 16
    idx = native.knn(k, part, part)                            # [B,2048,16], sorted by (distance, index)
    nb = torch.gather(part.unsqueeze(1).expand(-1, 2048, -1, -1), 2, idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
    dist = ((nb - part.unsqueeze(2)) ** 2).sum(-1)
    # (the kernel orders by the reference's expansion-form distance; direct differences agree to its rounding)
    assert bool((dist[..., 1:] >= dist[..., :-1] - 2e-6).all()) and bool((idx[..., 0] == torch.arange(2048, device="cuda")).all())
    sq = -2 * part @ part.transpose(1, 2) + (part ** 2).sum(-1)[:, :, None] + (part ** 2).sum(-1)[:, None, :]
    kth = sq.topk(k, dim=-1, largest=False).values.max(-1).values
    assert float((kth - dist[..., -1]).abs().max()) < 5e-6


def _same_gradient(a, b, exact):
    """Bitwise equal, or - for a gradient that came out of the grouped weight-gradient launch instead of the library's
    GEMM / the column-sum kernel (another summation order) - equal to rounding."""
    if exact:
        return torch.equal(a, b)
    return bool(torch.isfinite(b).all()) and (a - b).abs().max().item() <= 1e-4 * a.abs().max().item() + 1e-12


@pytest.mark.parametrize("grouped", [False, True])
def test_deferred_folds_match_immediate_ones(native, grouped, monkeypatch):
    """Bias / LayerNorm-weight gradients with their fold launches queued and run batched (`deferred_folds`) are bitwise
    the gradients of the immediate launches - also for a module applied twice in one graph (the engine adds its two
    gradients at once, so the queue is flushed at the second one) and for gradients nobody keeps.  `grouped`: the
    Linear weight (and bias) gradients are queued too and run as one grouped launch (`native.wgrad_defer`): same
    values to rounding, everything else still bitwise."""
    from imgadapointr_amd.models import Transformer_utils as TU
    monkeypatch.setattr(native, "GROUPED_WGRAD", grouped)
    torch.manual_seed(5)
    lin1, lin2 = TU.Linear(384, 768).cuda(), TU.Linear(768, 384).cuda()
    norm, act = TU.LayerNorm(384).cuda(), torch.nn.GELU()
    frozen = TU.LayerNorm(384).cuda().requires_grad_(False)
    x = torch.randn(8, 576, 384, device="cuda", requires_grad=True)
    params = list(lin1.parameters()) + list(lin2.parameters()) + list(norm.parameters())

    def run(defer):
        for p in params:
            p.grad = None
        h = frozen(x)
        for _ in range(2):                               # every module is applied twice
            s, y = TU.add_layer_norm(h, x, norm)
            h = lin2(TU.gelu(lin1(y), act)) + s
        loss = h.square().mean()
        with (native.deferred_folds() if defer else contextlib.nullcontext()):
            gx, = torch.autograd.grad(loss, x, retain_graph=True)
            loss.backward()
        assert not native._fold_keepalive and not native._wgrad_queue
        return [gx] + [p.grad.clone() for p in params]

    import contextlib
    exact = [True] + [not grouped] * 4 + [True, True]        # gx | lin1, lin2 weight + bias | norm weight + bias
    ref, got = run(False), run(True)
    for a, b, e in zip(ref, got, exact):
        assert _same_gradient(a, b, e)
    # once-applied modules: nothing forces a flush before the end of backward, and the result is still the same
    def run_once(defer):
        for p in params:
            p.grad = None
        s, y = TU.add_layer_norm(x, x, norm)
        loss = (lin2(TU.gelu(lin1(y), act)) + s).square().mean()
        with (native.deferred_folds() if defer else contextlib.nullcontext()):
            loss.backward()
        return [p.grad.clone() for p in params]

    for a, b, e in zip(run_once(False), run_once(True), exact[1:]):
        assert _same_gradient(a, b, e)


@pytest.mark.parametrize("grouped", [False, True])
def test_deferred_folds_guards(native, grouped, monkeypatch):
    """The failure modes the deferral could have, each with the queued outputs POISONED (NaN until the
    flush), so that a gradient adopted or added before its fold ran cannot pass:
      * a shared Linear / LayerNorm that is applied once through the deferring path (>= 2048 rows) and once through the
        plain aten path (few rows) in the same graph - the engine adds the two gradients at once;
      * `linear_on_broadcast_concat` on a parameter that also feeds a deferring Linear;
      * a stale `.grad` at the start of backward (an exception between backward and zero_grad): deferral is refused;
      * deferral is per stream: producers on another stream are not queued."""
    import contextlib
    from imgadapointr_amd.models import Transformer_utils as TU
    monkeypatch.setattr(native, "GROUPED_WGRAD", grouped)     # True: the Linear weight gradients are queued as well
    exact = [not grouped, not grouped, True, True]            # lin weight, bias | norm weight, bias
    torch.manual_seed(11)
    lin, norm = TU.Linear(384, 384).cuda(), TU.LayerNorm(384).cuda()
    big = torch.randn(8, 576, 384, device="cuda", requires_grad=True)        # 4608 rows: _LinearFn / _LayerNormFn
    small = torch.randn(2, 64, 384, device="cuda", requires_grad=True)       # 128 rows: F.linear / F.layer_norm
    g = torch.randn(8, 384, device="cuda")
    params = list(lin.parameters()) + list(norm.parameters())

    def graph():
        a = lin(norm(big))                                                   # deferring producers
        b = lin(norm(small))                                                 # plain paths, same parameters
        return a.square().mean() + b.square().mean()

    def run(defer, order):
        for p in params:
            p.grad = None
        loss = graph() if order == 0 else (lin(norm(small)).square().mean() + lin(norm(big)).square().mean())
        with (native.deferred_folds(params) if defer else contextlib.nullcontext()):
            loss.backward()
        assert not native._fold_keepalive and not native._wgrad_queue
        return [p.grad.clone() for p in params]

    old = native.FOLD_POISON
    native.FOLD_POISON = True
    try:
        for order in (0, 1):
            ref, got = run(False, order), run(True, order)
            for a, b, e in zip(ref, got, exact):
                assert torch.isfinite(b).all() and _same_gradient(a, b, e)
        # broadcast-concat Linear whose bias also belongs to a deferring call
        cat_lin = TU.Linear(384 + 384, 384).cuda()

        def run_cat(defer):
            cat_lin.zero_grad(set_to_none=True)
            y = TU.linear_on_broadcast_concat(cat_lin, g, big)               # plain path on cat_lin.bias
            z = cat_lin(torch.cat([big, big], -1))                           # deferring path on the same bias
            with (native.deferred_folds(list(cat_lin.parameters())) if defer else contextlib.nullcontext()):
                (y.square().mean() + z.square().mean()).backward()
            return [p.grad.clone() for p in cat_lin.parameters()]

        for a, b in zip(run_cat(False), run_cat(True)):
            assert torch.isfinite(b).all() and _same_gradient(a, b, not grouped)
        # stale .grad: accumulate-into-existing must not meet a queued buffer -> the context does not defer
        run(False, 0)
        stale = [p.grad.clone() for p in params]
        loss = graph()
        with native.deferred_folds(params):
            assert not native._fold_deferred
            loss.backward()
        accumulated = [p.grad.clone() for p in params]
        fresh = run(False, 0)
        for s0, acc, s1 in zip(stale, accumulated, fresh):
            assert torch.isfinite(acc).all() and torch.allclose(acc, s0 + s1, rtol=1e-5, atol=1e-7)
        # per-stream: deferral on the current stream does not capture a producer running on a side stream
        side = torch.cuda.Stream()
        xs = torch.randn(4096, 384, device="cuda")
        torch.cuda.synchronize()
        with native.deferred_folds():
            with torch.cuda.stream(side):
                cs = native.colsum(xs)
            side.synchronize()
            assert torch.isfinite(cs).all() and torch.allclose(cs, xs.sum(0), rtol=1e-4, atol=1e-3)
    finally:
        native.FOLD_POISON = old


def test_deferred_gradients_of_a_reshaped_leaf_weight(native):
    """`_linear_tokens` on a 1x1 convolution's weight viewed as a matrix (AdaPoinTr.increase_dim): the weight handed to the
    Linear node is a reshaped VIEW of the leaf, its queued gradient travels back through the view's backward (unsqueeze)
    and must be adopted unread - checked with poisoned outputs against the immediate path."""
    from imgadapointr_amd.models.AdaPoinTr import _linear_tokens
    torch.manual_seed(8)
    conv = torch.nn.Conv1d(384, 1024, 1).cuda()
    x = torch.randn(4608, 384, device="cuda", requires_grad=True)
    w = torch.randn(4608, 1024, device="cuda")
    res = {}
    old = native.FOLD_POISON
    native.FOLD_POISON = True
    try:
        for defer in (False, True):
            conv.zero_grad(set_to_none=True)
            x.grad = None
            y = _linear_tokens(x, conv.weight.squeeze(-1), conv.bias)
            import contextlib
            with (native.deferred_folds(list(conv.parameters())) if defer else contextlib.nullcontext()):
                (y * w).sum().backward()
                queued = len(native._wgrad_queue)
            assert queued == (1 if defer else 0)
            res[defer] = [x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()]
    finally:
        native.FOLD_POISON = old
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        assert a.shape == b.shape and _same_gradient(b, a, False)


def test_side_stream_fps_equals_the_plain_call(native):
    """`furthest_point_sample_async` with the side stream switched on returns the indices of the plain call, also when
    the input was produced on the current stream just before and the result is consumed right after (the two joins)."""
    from imgadapointr_amd.pointnet2_ops import pointnet2_utils as pu
    torch.manual_seed(2)
    base = torch.rand(8, 2048, 3, device="cuda")
    old = pu.OVERLAP_FPS
    pu.OVERLAP_FPS = True
    try:
        for _ in range(3):
            xyz = (base * 2.0 - 1.0).contiguous()                    # produced on the current stream
            h = pu.furthest_point_sample_async(xyz, 512)
            filler = (xyz @ xyz.transpose(1, 2)).sum()               # main-stream work underneath
            idx = h.result()
            picked = pu.gather_operation(xyz.transpose(1, 2).contiguous(), idx)      # consumed at once
            want = pu.furthest_point_sample(xyz, 512)
            assert torch.equal(idx, want) and torch.isfinite(filler) and picked.shape == (8, 3, 512)
            assert h.result() is idx                                 # idempotent
    finally:
        pu.OVERLAP_FPS = old


def test_clip_adamw_matches_torch(native):
    """ClipAdamW.step_clipped == clip_grad_norm_ + torch.optim.AdamW.step over several steps: two weight-decay groups,
    an lr change, a parameter without gradient on some steps, tensor sizes around the 4096-element chunk, clipping
    active and inactive; state keys stay torch's (checkpoint compatible)."""
    from imgadapointr_amd.utils.clip_adamw import ClipAdamW
    torch.manual_seed(3)
    shapes = [(384, 384), (384,), (1,), (4096,), (4097,), (1152, 384), (3, 5, 7), (8192 + 5,)]
    ref_p = [torch.randn(*s, device="cuda").requires_grad_(True) for s in shapes]
    our_p = [p.detach().clone().requires_grad_(True) for p in ref_p]

    def groups(ps):
        return [{"params": [p for p in ps if p.dim() == 1], "weight_decay": 0.0},
                {"params": [p for p in ps if p.dim() != 1], "weight_decay": 5e-2}]

    ref = torch.optim.AdamW(groups(ref_p), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    our = ClipAdamW(groups(our_p), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    for it in range(6):
        scale = 100.0 if it % 2 == 0 else 1e-3                    # clipping active / inactive (max_norm = 10)
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            if i == 2 and it in (1, 4):
                a.grad = b.grad = None                            # a parameter that is skipped now and then
                continue
            g = torch.randn_like(a) * scale
            a.grad, b.grad = g.clone(), g.clone()
        if it == 3:
            for o in (ref, our):
                for grp in o.param_groups:
                    grp["lr"] = 3e-3
        with_grad = [p for p in ref_p if p.grad is not None]
        want_norm = torch.nn.utils.clip_grad_norm_(with_grad, 10.0, norm_type=2)
        ref.step()
        got_norm = our.step_clipped(10.0)
        assert abs(float(got_norm) - float(want_norm)) <= 1e-5 * float(want_norm)
        for a, b in zip(ref_p, our_p):
            # a few ulp of the largest elements: the norm (hence the clip coefficient) is summed in another order
            tol = 8 * 2.0 ** -24 * float(a.detach().abs().max()) + 1e-7
            assert float((a.detach() - b.detach()).abs().max()) <= tol, (it, a.shape, float((a - b).abs().max()))
    our.state_dict()                                             # writes the step counters back into the state
    for a, b in zip(ref_p, our_p):
        sa, sb = ref.state[a], our.state[b]
        assert set(sa) == set(sb) == {"step", "exp_avg", "exp_avg_sq"} and float(sa["step"]) == float(sb["step"])
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=2e-5, atol=1e-6)
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-5, atol=1e-8)
    # a state_dict round trip through the stock class
    other = torch.optim.AdamW(groups([p.detach().clone().requires_grad_(True) for p in our_p]), lr=1e-2)
    other.load_state_dict(our.state_dict())
    # a NaN gradient: clip_grad_norm_ + step() of the reference turn every parameter into NaN at once (NaN norm -> NaN
    # coefficient); the fused step must not hide the blow-up behind a coefficient of 1
    for a, b in zip(ref_p, our_p):
        g = torch.randn_like(a)
        a.grad, b.grad = g.clone(), g.clone()
    ref_p[3].grad[7] = float("nan")
    our_p[3].grad[7] = float("nan")
    torch.nn.utils.clip_grad_norm_(ref_p, 10.0, norm_type=2)
    ref.step()
    assert torch.isnan(our.step_clipped(10.0))
    for a, b in zip(ref_p, our_p):
        assert torch.isnan(a).all() and torch.isnan(b).all()


def test_mlp_with_gelu_in_the_gemm_epilogue(native):
    """Mlp whose fc1 + GELU is one GEMM launch (GELU from the epilogue) against the unfused path: same outputs to fp32
    rounding, same gradients; and the epilogue GELU against float64."""
    from imgadapointr_amd.models import Transformer_utils as TU
    torch.manual_seed(9)
    mlp = TU.Mlp(384, 768).cuda()
    x = torch.randn(32, 576, 384, device="cuda", requires_grad=True)
    gy = torch.randn(32, 576, 384, device="cuda")
    res = {}
    for fused in (True, False):
        TU.FUSED_LINEAR_GELU = fused
        mlp.zero_grad()
        y = mlp(x)
        gx, = torch.autograd.grad(y, x, gy, retain_graph=True)
        y.backward(gy)
        res[fused] = [y.detach(), gx] + [p.grad.clone() for p in mlp.parameters()]
    TU.FUSED_LINEAR_GELU = True
    for a, b in zip(res[True], res[False]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * scale, float((a - b).abs().max()) / scale
    u, g = native.gemm_split3_gelu(x.detach().view(-1, 384), mlp.fc1.weight.detach(), mlp.fc1.bias.detach())
    ref_u = x.detach().view(-1, 384)[:2048].double() @ mlp.fc1.weight.detach().double().t() + mlp.fc1.bias.detach().double()
    ref_g = torch.nn.functional.gelu(ref_u)
    assert float((u[:2048].double() - ref_u).abs().max()) < 2e-6 * float(ref_u.abs().max())
    assert float((g[:2048].double() - ref_g).abs().max()) < 2e-6 * float(ref_g.abs().max())
    assert torch.equal(g, torch.nn.functional.gelu(u))          # the epilogue's GELU is aten's, bit for bit


def test_clip_adamw_survives_a_state_dict_round_trip(native):
    """load_state_dict replaces the moment tensors: the cached device table must follow them."""
    from imgadapointr_amd.utils.clip_adamw import ClipAdamW
    torch.manual_seed(4)
    a = [torch.randn(300, 40, device="cuda", requires_grad=True), torch.randn(40, device="cuda", requires_grad=True)]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    ref, our = torch.optim.AdamW(a, lr=1e-2), ClipAdamW(b, lr=1e-2)
    for it in range(4):
        for p, q in zip(a, b):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        torch.nn.utils.clip_grad_norm_(a, 1.0)
        ref.step()
        our.step_clipped(1.0)
        if it == 1:
            our.load_state_dict(our.state_dict())
    for p, q in zip(a, b):
        assert float((p - q).detach().abs().max()) <= 1e-6 * float(p.detach().abs().max()) + 1e-7


def test_clip_adamw_mixed_with_the_stock_step(native):
    """step_clipped, then the stock step() (what an external caller or an unsupported parameter set takes), then
    step_clipped again, against torch.optim.AdamW(fused=True): ClipAdamW defaults to torch's fused multi-tensor AdamW for
    its stock path, which reads `state["step"]` through a device pointer - so the counters this class creates must live
    where torch's own `_init_group` puts them."""
    from imgadapointr_amd.utils.clip_adamw import ClipAdamW
    torch.manual_seed(5)
    a = [torch.randn(300, 40, device="cuda", requires_grad=True), torch.randn(40, device="cuda", requires_grad=True),
         torch.randn(4097, device="cuda", requires_grad=True)]
    b = [p.detach().clone().requires_grad_(True) for p in a]
    ref, our = torch.optim.AdamW(a, lr=1e-2, fused=True), ClipAdamW(b, lr=1e-2)
    assert our.defaults["fused"]
    for it in range(6):
        for p, q in zip(a, b):
            g = torch.randn_like(p) * (30.0 if it % 2 else 0.1)
            p.grad, q.grad = g.clone(), g.clone()
        torch.nn.utils.clip_grad_norm_(a, 10.0)
        ref.step()
        if it in (1, 2, 4):                                       # the stock path: clip outside, plain step()
            torch.nn.utils.clip_grad_norm_(b, 10.0)
            our.step()
        else:
            our.step_clipped(10.0)
        if it == 0:
            assert all(our.state[q]["step"].is_cuda for q in b)
    our.state_dict()
    for p, q in zip(a, b):
        assert float(ref.state[p]["step"]) == float(our.state[q]["step"]) == 6.0
        assert float((p - q).detach().abs().max()) <= 8 * 2.0 ** -24 * float(p.detach().abs().max()) + 1e-7


def test_cross_attention_fused_kv_projection(native):
    """CrossAttention with k_map and v_map evaluated as ONE [2C, C] projection on the concatenated weights (packed
    [B,M,2,h,64] keys / values, packed gradient) against the two separate projections: outputs and every gradient
    (q, memory, all eight parameter tensors) to fp32 rounding, also under deferred folds with poisoned outputs."""
    from imgadapointr_amd.models import Transformer_utils as TU
    torch.manual_seed(21)
    B, N, M, C = 32, 576, 256, 384
    ca = TU.CrossAttention(C, C, num_heads=6, qkv_bias=True).cuda()
    for p in ca.parameters():
        torch.nn.init.normal_(p, std=0.05)
    q0 = torch.randn(B, N, C, device="cuda")
    v0 = torch.randn(B, M, C, device="cuda")
    w = torch.randn(B, N, C, device="cuda")
    res = {}
    old_poison = native.FOLD_POISON
    native.FOLD_POISON = True
    try:
        for fused in (True, False):
            TU.FUSED_KV_PROJECTION = fused
            ca.zero_grad(set_to_none=True)
            q, v = q0.clone().requires_grad_(True), v0.clone().requires_grad_(True)
            out = ca(q, v)
            with native.deferred_folds(list(ca.parameters())):
                (out * w).sum().backward()
            res[fused] = [out.detach(), q.grad, v.grad] + [p.grad.clone() for p in ca.parameters()]
    finally:
        TU.FUSED_KV_PROJECTION = True
        native.FOLD_POISON = old_poison
    names = ["out", "dq", "dv"] + [n for n, _ in ca.named_parameters()]
    for n, a, b in zip(names, res[True], res[False]):
        assert torch.isfinite(a).all(), n
        if "k_map.bias" in n:                      # zero in exact arithmetic (softmax shift invariance): rounding noise
            assert float(a.abs().max()) < 1e-3 * float(res[False][names.index("v_map.bias")].abs().max()) + 1e-4
            continue
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7, (n, float((a - b).abs().max()), scale)
