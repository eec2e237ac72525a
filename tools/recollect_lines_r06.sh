cd "$GRAFT_REPO_ROOT" || exit 2
for w in c2 c4; do
  NO_PLAN_AHEAD=1 tools/collect.sh r06_fbz_$w -s python3 tools/profile_fwd_bwd.py $w 30 zipf > /dev/null 2>&1
done
F=gpurun_out/r06_lines; mkdir -p $F
export NRX_BENCH_OUT=$F/bench_lines_zipf.jsonl; : > $NRX_BENCH_OUT
for z in c2 c4; do python3 bench.py --workload $z --ids zipf --steps 200 --warmup 20 --no-cpu-baseline > $F/bench_${z}_zipf.log 2>&1; done
export NRX_BENCH_OUT=$F/bench_lines.jsonl; : > $NRX_BENCH_OUT
for w in c2 c3 c4 c5; do python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > $F/bench_$w.log 2>&1; done
unset NRX_BENCH_OUT
python3 bench.py > $F/bench_c2_default.log 2>&1
grep -c metric $F/*.jsonl
