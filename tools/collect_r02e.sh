#!/bin/bash
# Round-2 final artifacts after the table-segmented planner sort and the settled-clock measurement of the matrix-core kernels:
# full GPU test run, smoke, the default bench command as the driver runs it, bench lines, fwd+bwd kernel stats, per-op timings,
# DCN-v2 forward kernel stats over a long (settled) run.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/r02_final_e; rm -rf $F; mkdir -p $F
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
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export NRX_BENCH_OUT=$F/bench_lines.jsonl
SECONDS=0; python3 bench.py > $F/bench_c2.log 2>&1; echo "default bench.py run: ${SECONDS} s"; tail -1 $F/bench_c2.log | cut -c1-200
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
{
for D in 320 512 112; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $F/dcn2f_$D -- python3 tools/run_dcn2.py $D 600 > $F/dcn2f_$D.log 2>&1
  echo "== DCN-v2 layer forward (inference form, layer 0: x0 == x_l), B = 65536, D = $D, 600 back-to-back launches (kernel average incl. the first ~40 ms of clock ramp)"
  stats $F/dcn2f_$D
done
} > $F/dcn_v2_fwd_settled_kernel_stats.txt 2>&1
python3 tools/bench_ops.py > $F/bench_ops.log 2>&1
{ python3 tools/probe_dcn1_bwd.py 16 4096 65536; SEP=1 python3 tools/probe_dcn1_bwd.py 65536; python3 tools/probe_small_bwd.py; for w in c2 c4 c5; do python3 tools/profile_plan.py $w 300 uniform; NRX_PLAN_SORT=rocprim python3 tools/profile_plan.py $w 300 uniform; done; } 2>&1 | grep "us" > $F/direct_kernel_timings.txt
grep -v "amdgpu.ids\|Warning" $F/bench_ops.log | tail -40
