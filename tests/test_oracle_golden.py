"""Pins the CPU oracle: against the golden vectors produced by the reference's own pure-torch
arithmetic (square_distance / knn_point), and against float64 brute force for the operators the
reference only ships as CUDA (Chamfer) or not at all (pointnet2_ops FPS)."""
import numpy as np

import parity_checks as pc


def test_square_distance_vs_reference(ref):
    g = pc.load("functional.npz")
    got = ref.square_distance(g["new_xyz"], g["xyz"])
    # same formula; the reference's BLAS may round the K=3 dot product differently (last ulp)
    assert np.max(np.abs(got - g["square_distance"])) < 2e-6
    assert np.mean(got == g["square_distance"]) > 0.5


def test_knn_vs_reference(ref):
    g = pc.load("functional.npz")
    got = np.sort(ref.knn(8, g["xyz"], g["new_xyz"]), -1)
    want = g["knn_sorted"]
    sq = g["square_distance"]
    bad = 0
    for b in range(got.shape[0]):
        for s in range(got.shape[1]):
            if not np.array_equal(got[b, s], want[b, s]):
                diff = set(got[b, s]) ^ set(want[b, s])
                ds = [sq[b, s, j] for j in diff]
                assert max(ds) - min(ds) < 1e-6  # only a rounding-level tie may differ
                bad += 1
    assert bad <= 2


def test_knn_order_and_ties(ref):
    rng = np.random.default_rng(0)
    base = rng.random((1, 40, 3), dtype=np.float32)
    c = np.concatenate([base, base, base], 1)  # every point three times -> exact ties
    q = base[:, :10]
    idx = ref.knn(6, c, q)
    d = ref.square_distance(q, c)
    for s in range(10):
        row = idx[0, s]
        keys = [(d[0, s, j], j) for j in row]
        assert keys == sorted(keys)                       # (distance, index) ascending
        assert list(row[:3]) == [s, s + 40, s + 80]       # the three copies of itself, by index
        order = sorted(range(120), key=lambda j: (d[0, s, j], j))[:6]
        assert list(row) == order


def test_topk_rows_is_torch_topk_with_a_defined_tie_order(ref):
    """GDANet's feature-space kNN is `pairwise.topk(k, dim=-1)[1]` (reference GDANet_util.py:5-11): the restatement
    returns torch.topk's indices wherever the values differ, and (value desc, column asc) among equal values."""
    import torch
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 37, 512)).astype(np.float32)
    for k in (1, 30, 64, 512):
        assert np.array_equal(ref.topk_rows(x, k), torch.from_numpy(x).topk(k, dim=-1)[1].numpy())
    y = (np.round(x * 3) / 3).astype(np.float32)          # many exact ties
    y[0, 0, :5] = [0.0, -0.0, np.inf, -np.inf, 0.0]
    idx = ref.topk_rows(y, 100)
    val = np.take_along_axis(y, idx, -1)
    assert np.array_equal(val, torch.from_numpy(y).topk(100, dim=-1)[0].numpy())      # the same multiset, sorted
    step = np.diff(val, axis=-1)
    assert (step <= 0).all() and (np.diff(idx, axis=-1)[step == 0] > 0).all()
    assert idx[0, 0, 0] == 2 and idx.dtype == np.int64


def test_chamfer_vs_float64_bruteforce(ref):
    rng = np.random.default_rng(1)
    a = (rng.random((3, 257, 3), dtype=np.float32) - 0.5)
    b = (rng.random((3, 403, 3), dtype=np.float32) - 0.5)
    d1, d2, i1, i2 = ref.chamfer_forward(a, b)
    D = ((a[:, :, None, :].astype(np.float64) - b[:, None, :, :].astype(np.float64)) ** 2).sum(-1)
    assert np.max(np.abs(d1 - D.min(2))) < 1e-7 and np.max(np.abs(d2 - D.min(1))) < 1e-7
    assert np.array_equal(i1, D.argmin(2)) and np.array_equal(i2, D.argmin(1))


def test_chamfer_tie_rule_and_identity(ref):
    a = np.zeros((1, 5, 3), np.float32)
    b = np.tile(np.array([[1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, 0, 1]], np.float32), (1, 2, 1))
    _, _, i1, _ = ref.chamfer_forward(a, b)   # all eight candidates at distance 1: index 0 wins
    assert (i1 == 0).all()
    x = np.random.default_rng(2).random((2, 100, 3), dtype=np.float32)
    d1, d2, i1, i2 = ref.chamfer_forward(x, x)
    assert (d1 == 0).all() and (d2 == 0).all()
    assert (i1 == np.arange(100)).all() and (i2 == np.arange(100)).all()


def test_chamfer_backward_is_the_gradient(ref):
    """Finite differences of sum(w1*dist1) + sum(w2*dist2) in float64 on the oracle's own forward."""
    rng = np.random.default_rng(3)
    a = (rng.random((1, 6, 3)) - 0.5).astype(np.float32)
    b = (rng.random((1, 9, 3)) - 0.5).astype(np.float32)
    w1 = rng.standard_normal((1, 6)).astype(np.float32)
    w2 = rng.standard_normal((1, 9)).astype(np.float32)
    _, _, i1, i2 = ref.chamfer_forward(a, b)
    ga, gb = ref.chamfer_backward(a, b, i1, i2, w1, w2)

    def f(a_, b_):
        D = ((a_[:, :, None, :] - b_[:, None, :, :]) ** 2).sum(-1)
        return (w1 * D.min(2)).sum() + (w2 * D.min(1)).sum()

    eps = 1e-6
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    for arr, grad, which in ((a64, ga, 0), (b64, gb, 1)):
        for j in range(arr.shape[1]):
            for d in range(3):
                p = arr.copy(); p[0, j, d] += eps
                m = arr.copy(); m[0, j, d] -= eps
                num = (f(p, b64) - f(m, b64)) / (2 * eps) if which == 0 else (f(a64, p) - f(a64, m)) / (2 * eps)
                assert abs(num - grad[0, j, d]) < 1e-4


def test_fps_definition(ref):
    rng = np.random.default_rng(4)
    x = (rng.random((2, 300, 3), dtype=np.float32) - 0.5)
    x[:, 250:] = 0.0                                  # zero padding: never selected
    idx = ref.furthest_point_sample(x, 40)
    assert (idx[:, 0] == 0).all() and (idx < 250).all()
    for b in range(2):
        assert len(set(idx[b])) == 40
        # replay in float64: each pick maximises the distance to the picks so far
        pts = x[b].astype(np.float64)
        mind = np.full(300, 1e10)
        valid = (pts ** 2).sum(-1) > 1e-3
        last = 0
        for j in range(1, 40):
            mind = np.minimum(mind, ((pts - pts[last]) ** 2).sum(-1))
            cand = np.where(valid, mind, -1.0)
            assert abs(cand[idx[b, j]] - cand.max()) < 1e-6
            last = idx[b, j]
    assert np.array_equal(ref.furthest_point_sample(x, 17), idx[:, :17])   # prefix-consistent
