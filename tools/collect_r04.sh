#!/bin/bash
# Round-4 judged artifacts (run through gpurun): counters for every bench workload's headline launch and for the forward + backward legs,
# bench lines of every workload (+ Zipf ids), the sharded engine at world 1.   tools/collect_r04.sh [profiles|lines|all]
cd "$GRAFT_REPO_ROOT" || exit 2
what=${1:-all}
if [ $what = profiles ] || [ $what = all ]; then
  for w in c2 c3 c4 c5; do
    tools/collect.sh r04_$w python3 bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --headline-only > /dev/null 2>&1
  done
  for w in c2 c4 c5; do
    NO_PLAN_AHEAD=1 tools/collect.sh r04_fb_$w python3 tools/profile_fwd_bwd.py $w 30 uniform > /dev/null 2>&1
  done
fi
if [ $what = lines ] || [ $what = all ]; then
  F=gpurun_out/r04_lines; rm -rf $F; mkdir -p $F
  export NRX_BENCH_OUT=$F/bench_lines.jsonl
  SECONDS=0; python3 bench.py > $F/bench_c2.log 2>&1; echo "default bench.py run: ${SECONDS} s"
  for w in c3 c4 c5; do python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_$w.log 2>&1; done
  export NRX_BENCH_OUT=$F/bench_lines_zipf.jsonl
  for z in c2 c4; do python3 bench.py --workload $z --ids zipf --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_${z}_zipf.log 2>&1; done
  unset NRX_BENCH_OUT
  tools/sharded_world1.sh > $F/sharded_world1.log 2>&1
fi
