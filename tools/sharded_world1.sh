#!/bin/bash
# Dev: the sharded engine at world 1 (the only place it can be measured here), row layout: bench lines with the bound training-step leg.
#   tools/sharded_world1.sh [workloads...]      (default: c2 c5 c3 c4)
cd "$GRAFT_REPO_ROOT" || exit 2
F=gpurun_out/sharded_w1; mkdir -p $F
export NRX_BENCH_OUT=$F/lines.jsonl; : > $NRX_BENCH_OUT
for w in ${@:-c2 c5 c3 c4}; do
  python3 bench.py --workload $w --force-sharded --shard-mode row --steps 100 --warmup 10 --no-cpu-baseline > $F/$w.log 2>&1
  python3 - <<PY
import json
for l in open("$F/lines.jsonl"):
    d = json.loads(l)
    if d["config"]["workload"].startswith("$w:"):
        fb = d.get("fwd_bwd", {})
        print("$w", "fwd ms", round(d["ms_per_step"], 4), "engine", d["config"].get("layout", {}).get("engine"), "fwd_bwd", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in fb.items() if k in ("ms_per_step", "gpu_ms_per_step_rank0", "host_bound", "error", "skipped")}, "frac", round(fb.get("roofline", {}).get("frac", 0), 3))
PY
done
