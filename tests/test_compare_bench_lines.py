"""tools/compare_bench_lines.py -- the regression gate over the committed bench lines (round 5 lost 47-64 % on the Zipf legs and nothing
compared a round's lines with the previous round's).  CPU only: it reads profiles/*.jsonl."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "compare_bench_lines.py")


def _run(*args):
    return subprocess.run([sys.executable, TOOL, *args], capture_output=True, text=True, timeout=120)


def test_gate_flags_round_5_against_round_4():
    """The committed history holds the regression: round 5's Zipf forward + backward legs are 1.47x / 1.64x round 4's."""
    r = _run("5", "4")
    assert r.returncode == 1, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "SLOWER" in l]
    assert any("zipf" in l and "fwd_bwd.ms_per_step" in l and l.split()[0] == "c2" for l in lines), r.stdout
    assert any("zipf" in l and "fwd_bwd.ms_per_step" in l and l.split()[0] == "c4" for l in lines), r.stdout
    assert not any("host" in l for l in lines)             # host-time legs are listed, never gated


def test_gate_passes_equal_lines_and_fails_a_slower_leg(tmp_path):
    base = {"metric": "m", "n_gpus": 1, "ms_per_step": 0.06, "config": {"workload": "c2: x"},
            "fwd_bwd": {"ms_per_step": 0.19, "host_us_per_step": 100.0}, "roofline": {"kernel_ms_mean": 0.0599}}
    a, b = tmp_path / "new.jsonl", tmp_path / "old.jsonl"
    b.write_text(json.dumps(base) + "\n")
    a.write_text(json.dumps(base) + "\n")
    assert _run(str(a), str(b)).returncode == 0
    worse = json.loads(json.dumps(base))
    worse["fwd_bwd"]["host_us_per_step"] = 300.0          # host time only: listed, not gated
    a.write_text(json.dumps(worse) + "\n")
    r = _run(str(a), str(b))
    assert r.returncode == 0 and "not gated" in r.stdout
    worse["fwd_bwd"]["ms_per_step"] = 0.22                # + 16 %
    a.write_text(json.dumps(worse) + "\n")
    r = _run(str(a), str(b))
    assert r.returncode == 1 and "fwd_bwd.ms_per_step" in r.stdout
    assert _run(str(a), str(b), "--tol=0.2").returncode == 0
