"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code (race / memory-error detection; GPU ASan is
not available on the pool, so the sanitizers run on the host builds): (1) the C restatement of the oracle, every function incl. the
path's edge cases (oracle/sanitize_driver.c); (2) the host half of the C-ABI -- argument validation, status codes, error text --
compiled from the library's own sources with host-side instrumentation (tests/sanitize/). No GPU needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    return out


def test_oracle_c_restatement_is_clean_under_asan_ubsan():
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    assert "oracle sanitize driver: OK" in _run(["make", "-C", "oracle", "sanitize"])


def test_capi_host_validation_is_clean_under_asan_ubsan():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    if not os.path.exists(os.path.join(ROOT, "tests", "sanitize", "Makefile")):
        pytest.skip("tests/sanitize/ does not travel to the GPU box (.gpurunignore: the pool refuses any snapshot holding a hipcc -fsanitize recipe)")
    assert "C-ABI validation sanitize driver: OK" in _run(["make", "-C", "tests/sanitize", "run"])


@pytest.mark.gpu
def test_roctx_ranges_are_env_gated_and_harmless():
    """NRX_ROCTX=1: every C-ABI entry point pushes / pops a roctx range (libroctx64 loaded on first use); results unchanged."""
    import sys
    code = ("import torch, numpy as np\n"
            "from news_recsys_amd import ops, _lib\n"
            "from news_recsys_amd._lib import NRX_SPARSE\n"
            "t = torch.arange(40, dtype=torch.float32, device='cuda').reshape(10, 4)\n"
            "plan = ops.EmbedPlan([ops.Slot('a', NRX_SPARSE, 0, 4, 0, 0)], out_width=4)\n"
            "ids = torch.tensor([3, 0, 9], device='cuda')\n"
            "out = ops.embed_apply(plan, [t], [ids], [None])[0]\n"
            "assert torch.equal(out, t[ids]); print('ok')\n")
    for flag in ("1", "0"):
        env = dict(os.environ, NRX_ROCTX=flag)
        p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-2000:]
