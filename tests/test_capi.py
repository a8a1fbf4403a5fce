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
