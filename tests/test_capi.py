"""The C-ABI library loads and exports every symbol include/ptops.h declares (no compute here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptops_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from imgadapointr_amd import _native
    lib = _native.load()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ptops.h but not exported"
    assert set(names) == set(_native.EXPORTS)
    assert lib.ptops_abi_version() == 1
    assert isinstance(lib.ptops_last_error(), bytes)


def test_workspace_query_is_pure_host_code():
    from imgadapointr_amd import _native
    lib = _native.load()
    # [sweep: sorted float4 copies of both clouds | MFMA filter: bf16 operand tiles of both clouds,
    #  packed minima | brute force: packed minima when it splits its candidates], each part rounded up to 256 bytes
    def a256(v):
        return (v + 255) // 256 * 256

    def mfma(B, n, m):      # 48 bytes of bf16 operand pieces per candidate, candidates padded to 256-point superblocks
        pad = lambda k: (k + 255) // 256 * 256
        return a256(a256(48 * B * pad(n)) + a256(48 * B * pad(m)) + 8 * B * (n + m))

    assert lib.ptops_chamfer_sweep_available(32, 16384, 16384) == 1
    assert lib.ptops_chamfer_forward_workspace_bytes(32, 16384, 16384) == \
        16 * 32 * 32768 + mfma(32, 16384, 16384) + 8 * 32 * 32768
    assert lib.ptops_chamfer_sweep_available(32, 512, 16384) == 0
    assert lib.ptops_chamfer_forward_workspace_bytes(32, 512, 16384) == mfma(32, 512, 16384) + 8 * 32 * (512 + 16384)
    assert lib.ptops_chamfer_forward_workspace_bytes(1, 100, 77) == mfma(1, 100, 77)      # ragged sizes pad to 256
    assert lib.ptops_chamfer_forward_workspace_bytes(0, 10, 10) == 0


def test_wgrad_group_plan_is_a_partition_of_the_work():
    """`ptops_wgrad_group_plan` (pure host code): every (problem, contraction split, tile) is exactly one job, the splits of
    a problem cover its k-steps once, all tiles of a (problem, split) sit in ONE XCD queue (job index mod 8), the queues
    are balanced, every output element has exactly one reduce job, and the workspace ranges do not overlap."""
    import numpy as np
    from imgadapointr_amd import _native
    lib = _native.load()
    mnk = np.array([[18432, 384, 384]] * 9 + [[18432, 1152, 384]] * 3 + [[8192, 384, 768]] * 4 + [[2048, 200, 132], [16, 128, 128],
                   [4624, 96, 512]], dtype=np.int32)
    want_db = np.array([i % 2 for i in range(len(mnk))], dtype=np.uint8)
    n = len(mnk)
    for target in (0, 24, 100, 100000):
        need = lib.ptops_wgrad_group_plan(mnk.ctypes.data, want_db.ctypes.data, n, target, None, 0)
        assert need > 0
        blob = np.zeros(need, dtype=np.uint8)
        assert lib.ptops_wgrad_group_plan(mnk.ctypes.data, want_db.ctypes.data, n, target, blob.ctypes.data, need) == need
        head = blob[:16].view(np.int32)
        n_jobs, n_rjobs = int(head[2]), int(head[3])
        assert int(head[1]) == n and n_jobs % 8 == 0
        ws_bytes = int(blob[16:24].view(np.int64)[0])
        assert ws_bytes == lib.ptops_wgrad_group_workspace_bytes(blob.ctypes.data)
        pp = blob[32:32 + 32 * n]
        part_base = pp.view(np.int64).reshape(n, 4)[:, 0]
        bpart_base = pp.view(np.int64).reshape(n, 4)[:, 1]
        ints = pp.view(np.int32).reshape(n, 8)
        splits, per, tiles_k = ints[:, 4], ints[:, 5], ints[:, 6]
        jobs = blob[32 + 32 * n:32 + 32 * n + 16 * n_jobs].view(np.int32).reshape(n_jobs, 4)
        rjobs = blob[32 + 32 * n + 16 * n_jobs:].view(np.int32).reshape(n_rjobs, 4)
        seen, queue_of, load = set(), {}, np.zeros(8)
        for j, (p, tile, split, _) in enumerate(jobs):
            if p < 0:
                continue
            M, N, K = mnk[p]
            tiles = -(-N // 128) * -(-K // 128)
            assert 0 <= tile < tiles and 0 <= split < splits[p] and tiles_k[p] == -(-K // 128)
            assert (p, tile, split) not in seen
            seen.add((p, tile, split))
            assert queue_of.setdefault((p, split), j % 8) == j % 8
            load[j % 8] += min(per[p], M // 16 - split * per[p])
        for p, (M, N, K) in enumerate(mnk):
            tiles = -(-N // 128) * -(-K // 128)
            assert all((p, t, s) in seen for t in range(tiles) for s in range(splits[p]))
            assert (splits[p] - 1) * per[p] < M // 16 <= splits[p] * per[p]          # no empty split, full cover
        assert len(seen) == sum(-(-N // 128) * -(-K // 128) * splits[p] for p, (M, N, K) in enumerate(mnk))
        assert load.max() - load.min() <= max(per.max() * 27, 0.2 * load.mean())     # within one (largest) group
        # reduce jobs: 1024-float chunks of every dW, and of db where asked
        for p, (M, N, K) in enumerate(mnk):
            chunks = sorted(c for q, kind, c, _ in rjobs if q == p and kind == 0)
            assert chunks == list(range(-(-N * K // 1024)))
            chunks = sorted(c for q, kind, c, _ in rjobs if q == p and kind == 1)
            assert chunks == (list(range(-(-N // 1024))) if want_db[p] else [])
        # workspace ranges (floats) in order, inside the workspace
        end = 0
        for p, (M, N, K) in enumerate(mnk):
            assert part_base[p] >= end and part_base[p] % 4 == 0
            end = part_base[p] + splits[p] * N * K
            if want_db[p]:
                assert bpart_base[p] >= end and bpart_base[p] % 4 == 0
                end = bpart_base[p] + splits[p] * N
        assert end * 4 <= ws_bytes
    assert lib.ptops_wgrad_group_plan(np.array([[100, 8, 8]], dtype=np.int32).ctypes.data, None, 1, 0, None, 0) == -1
    assert lib.ptops_wgrad_group_plan(np.array([[128, 6, 8]], dtype=np.int32).ctypes.data,
                                      np.array([1], dtype=np.uint8).ctypes.data, 1, 0, None, 0) == -1        # db needs N % 4
    assert lib.ptops_wgrad_problem_bytes() == 72


def test_no_cpu_fallback():
    import pytest
    import torch
    from imgadapointr_amd import _native
    assert _native.backend_for_testing() is None
    with pytest.raises(_native.PtopsError):
        _native.chamfer_forward(torch.rand(1, 8, 3), torch.rand(1, 8, 3))
    with pytest.raises(_native.PtopsError):
        _native.knn(4, torch.rand(1, 8, 3), torch.rand(1, 2, 3))


def test_oracle_is_not_imported_by_the_package():
    import subprocess
    import sys
    code = ("import sys; import imgadapointr_amd.models, imgadapointr_amd.tools.runner, imgadapointr_amd.utils.metrics;"
            "bad=[m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; assert not bad, bad")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "imgadapointr_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libptops_ref" not in src, fn


def test_chamfer_policy_ladder_on_cpu():
    """Host logic of the adaptive Chamfer choice (_native._ChamferPolicy): sweep -> MFMA filter -> brute force, demoted by
    the device counters of an EARLIER call, re-probed after `hold` calls.  No GPU: the pending read-back is faked."""
    from imgadapointr_amd import _native as nat

    class Done:
        def query(self):
            return True

    class Stats:
        def __init__(self, vals):
            self.vals = vals

        def tolist(self):
            return self.vals

    pol = nat._ChamferPolicy()
    big = 1 << 33
    # sweep-eligible, large: starts with the sweep in both directions
    algos, st = pol.choose("k", True, big)
    assert algos == [nat.CHAMFER_SWEEP, nat.CHAMFER_SWEEP]
    # direction 1 evaluated 90 % of its chunks: demoted to the MFMA filter, direction 0 (5 %) keeps sweeping
    st["pending"] = (Stats([5, 100, 90, 100]), Done(), tuple(algos))
    algos, st = pol.choose("k", True, big)
    assert algos == [nat.CHAMFER_SWEEP, nat.CHAMFER_MFMA] and st["left"][1] == pol.hold - 1
    # the filter drowns in exact ties (10 % of the superblocks re-evaluated): down to brute force
    st["pending"] = (Stats([0, 0, 10, 100]), Done(), tuple(algos))
    algos, st = pol.choose("k", True, big)
    assert algos == [nat.CHAMFER_SWEEP, nat.CHAMFER_BRUTE]
    for _ in range(pol.hold):                       # after `hold` calls the preferred algorithm is probed again
        algos, st = pol.choose("k", True, big)
    assert algos == [nat.CHAMFER_SWEEP, nat.CHAMFER_SWEEP]
    # small launches never use the filter; without the sweep the ladder is MFMA -> brute or just brute
    assert pol.choose("s", False, 1 << 20)[0] == [nat.CHAMFER_BRUTE, nat.CHAMFER_BRUTE]
    assert pol.choose("m", False, big)[0] == [nat.CHAMFER_MFMA, nat.CHAMFER_MFMA]
    assert pol.choose("t", True, 1 << 28)[0] == [nat.CHAMFER_SWEEP, nat.CHAMFER_SWEEP]
    _, st = pol.choose("t", True, 1 << 28)
    st["pending"] = (Stats([80, 100, 80, 100]), Done(), (nat.CHAMFER_SWEEP, nat.CHAMFER_SWEEP))
    assert pol.choose("t", True, 1 << 28)[0] == [nat.CHAMFER_BRUTE, nat.CHAMFER_BRUTE]
