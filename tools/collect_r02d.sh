#!/bin/bash
# Round-2 artifacts of the DCN-v2 backward + per-op timings (re-collected after the epilogue / fold / edge-skip changes), full GPU
# test run, smoke, and the default bench command as the driver runs it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/r02_final_d; rm -rf $F; mkdir -p $F
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
SECONDS=0; python3 bench.py > $F/bench_default.log 2>&1; echo "default bench.py run: ${SECONDS} s"; tail -1 $F/bench_default.log | cut -c1-300
{
for D in 320 112; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $F/dcn2_$D -- python3 tools/profile_dcn2_bwd.py $D > $F/dcn2_$D.log 2>&1
  echo "== DCN-v2 layer forward + backward, B = 65536, D = $D (per-kernel averages)"
  stats $F/dcn2_$D
done
} > $F/dcn_v2_bwd_kernel_stats.txt 2>&1
python3 tools/bench_ops.py > $F/bench_ops.log 2>&1
grep "dcn_v2" $F/bench_ops.log
