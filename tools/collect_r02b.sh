#!/bin/bash
# Round-2 artifacts that depend on the backward kernels (re-collected after the bag fast path); see tools/collect_r02.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/r02_final_b; rm -rf $F; mkdir -p $F
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
export NRX_BENCH_OUT=$F/bench_lines.jsonl
python3 bench.py > $F/bench_c2.log 2>&1
for w in c3 c4 c5; do python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_$w.log 2>&1; done
unset NRX_BENCH_OUT
export NRX_BENCH_OUT=$F/bench_lines_zipf.jsonl
for z in c2 c4; do python3 bench.py --workload $z --ids zipf --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_${z}_zipf.log 2>&1; done
unset NRX_BENCH_OUT
{
for w in "c2 uniform" "c2 zipf" "c4 uniform" "c4 zipf" "c5 uniform"; do
  set -- $w
  rocprofv3 --kernel-trace --stats --output-format csv -d $F/fb_$1$2 -- python3 tools/profile_fwd_bwd.py $1 30 $2 > $F/fb_$1$2.log 2>&1
  echo "== forward (training form) + row-sparse backward, workload $1, $2 ids (30 warm-up + 30 timed steps; per-kernel averages)"
  grep "fwd+bwd" $F/fb_$1$2.log
  stats $F/fb_$1$2
done
} > $F/fwd_bwd_kernel_stats.txt 2>&1
python3 tools/bench_ops.py > $F/bench_ops.log 2>&1
ls $F
