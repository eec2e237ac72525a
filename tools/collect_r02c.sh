#!/bin/bash
# Round-2 artifacts of the row-sharded engine at world = 1 (re-collected after the ring owner gather and the local / exchange overlap)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/r02_final_c; rm -rf $F; mkdir -p $F
stats() {
python3 - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "distribution" in n or "FillFunctor" in n or float(r["AverageNs"]) < 3000: continue
        print(f'{float(r["AverageNs"]) / 1e3:9.1f} us x{r["Calls"]:>5}  {n[:120]}')
PY
}
python -m pytest tests -x -q -m gpu > $F/pytest_gpu.log 2>&1; tail -2 $F/pytest_gpu.log
export NRX_BENCH_OUT=$F/bench_lines_sharded_world1.jsonl
for z in c2 c4; do python3 bench.py --workload $z --force-sharded --shard-mode row --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_${z}_sh1.log 2>&1; done
python3 bench.py --workload c5 --force-sharded --shard-mode auto --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_c5_auto.log 2>&1
unset NRX_BENCH_OUT
{
for z in "c2 row" "c4 row" "c5 auto"; do
  set -- $z
  rocprofv3 --kernel-trace --stats --output-format csv -d $F/sh1_$1 -- python3 bench.py --workload $1 --force-sharded --shard-mode $2 --steps 50 --warmup 10 --no-cpu-baseline --headline-only > /dev/null 2>&1
  echo "== row-sharded engine at world = 1, workload $1, layout $2 (per-kernel averages of one step's launches)"
  stats $F/sh1_$1
done
} > $F/sharded_world1_kernel_stats.txt 2>&1
cat $F/sharded_world1_kernel_stats.txt | cut -c1-150
python3 - $F/bench_lines_sharded_world1.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["config"]["workload"][:100], "|", round(d["ms_per_step"] * 1e3, 1), "us")
PY
