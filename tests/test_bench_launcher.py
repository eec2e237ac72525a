"""bench.py's own N > 1 launcher (`python bench.py --gpus N` without torchrun): rank environment, relay of rank 0's
line, worst return code, failure and stall handling -- with stand-in rank programs on the CPU -- and, on the GPU box,
the real thing: `python bench.py --gpus 2` end to end over the host-staged test transport (both ranks on cuda:0)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD_OK = r"""
import json, os, sys
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
print("noise from rank", r)
if r == 0:
    print(json.dumps({"n_gpus": w, "value": 1.0, "args": sys.argv[1:]}))
"""
CHILD_FAIL = r"""
import os, sys, time
if os.environ["RANK"] == "1":
    sys.exit(7)
time.sleep(600)
"""
CHILD_HANG = "import time; time.sleep(600)"


def test_run_ranks_sets_rank_env_and_relays_rank0():
    import bench
    rc, lines, note = bench.run_ranks(3, [sys.executable, "-c", CHILD_OK, "--x"], 60.0)
    assert rc == 0 and note is None
    js = [json.loads(l) for l in lines if l.startswith("{")]
    assert len(js) == 1 and js[0]["n_gpus"] == 3 and js[0]["args"] == ["--x"]
    assert not any("rank 1" in l or "rank 2" in l for l in lines)       # other ranks' stdout is not relayed


def test_run_ranks_failed_rank_takes_the_job_down(monkeypatch):
    import bench
    import time
    t0 = time.monotonic()
    rc, lines, note = bench.run_ranks(2, [sys.executable, "-c", CHILD_FAIL], 120.0)
    assert rc != 0 and "rank 1" in note
    assert time.monotonic() - t0 < 60


def test_run_ranks_stalled_job_is_killed_and_nonzero():
    import bench
    rc, lines, note = bench.run_ranks(2, [sys.executable, "-c", CHILD_HANG], 1.5)
    assert rc != 0 and "did not finish" in note


def test_plain_python_gpus2_without_gpu_fails_loudly_not_silently():
    """No GPU here: the launcher must start the ranks, see them fail and exit non-zero without printing a JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["NRX_BENCH_LAUNCH_TIMEOUT"] = "120"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert not any(l.startswith('{"metric"') for l in p.stdout.splitlines())


@pytest.mark.gpu
def test_bench_gpus2_self_launched_host_staged_end_to_end():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"NRX_BENCH_HOST_STAGED": "1", "NRX_BENCH_LAUNCH_TIMEOUT": "900"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--secondary-timeout", "400"], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    js = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(js) == 1, p.stdout[-2000:]
    line = js[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and line["scaling"] == "weak"
    assert "launcher" in line
    assert "strong_scaling" in line or "secondary_note" in line
    # the N > 1 line describes itself (round 6): the headline layout is row-sharded for every workload -- the default c2 here: 26 tables, none
    # replicated --, the process group reports its own size, the roofline is per rank, the ratio to the direct single-GPU path is stated
    lay = line["config"]["layout"]
    assert lay["mode"] == "row" and lay["tables_row_sharded"] == 26 and lay["tables_replicated"] == 0 and lay["engine"] == "feat"
    assert line["rccl_ranks"]["world_size"] == 2 and "gloo" in line["rccl_ranks"]["backend"] and "NOT a measurement" in line["rccl_ranks"]["transport"]
    assert line["roofline"]["achieved"] > 0 and line["roofline"]["algorithmic_bytes_per_launch"] == 3540 * 65536
    assert line["scaling_vs_1gpu"]["direct_1gpu_value"] > 0 and "profiles/" in line["scaling_vs_1gpu"]["direct_1gpu_source"]
    if "secondary_note" not in line:
        assert line["a2a"]["GBps_per_link"] > 0 and line["a2a"]["xgmi_links_per_gpu"] == 7
        assert line["other_layout"]["layout_mode"] == "auto" and line["other_layout"]["tables_replicated"] == 26
        fb = line["fwd_bwd"]
        assert fb["engine"] == "feat" and fb["roofline"]["frac"] > 0 and "PreparedShardedStep" in fb["mode"]


@pytest.mark.gpu
def test_bench_single_rank_rccl_forced_dist():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"NRX_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517", "RANK": "0", "WORLD_SIZE": "1",
                "LOCAL_RANK": "0"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--force-sharded",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    js = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(js) == 1 and js[0]["n_gpus"] == 1
