#!/bin/bash
# Round-5 judged artifacts (run through gpurun): counters for the headline launch and for the forward + backward legs (c2 with the one-kernel
# planner, c4 / c5 with the sorted planners), bench lines of every workload with >= 100 timed steps (+ Zipf ids), the sharded engine at world 1
# with its training-step leg.   tools/collect_r05.sh [profiles|lines|all]
cd "$GRAFT_REPO_ROOT" || exit 2
what=${1:-all}
if [ $what = profiles ] || [ $what = all ]; then
  tools/collect.sh r05_c2 python3 bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline --headline-only > /dev/null 2>&1
  for w in c2 c4 c5; do
    NO_PLAN_AHEAD=1 tools/collect.sh r05_fb_$w python3 tools/profile_fwd_bwd.py $w 30 uniform > /dev/null 2>&1
  done
fi
if [ $what = lines ] || [ $what = all ]; then
  F=gpurun_out/r05_lines; rm -rf $F; mkdir -p $F
  export NRX_BENCH_OUT=$F/bench_lines.jsonl
  SECONDS=0; python3 bench.py > $F/bench_c2_default.log 2>&1; echo "default bench.py run: ${SECONDS} s"
  : > $F/bench_lines.jsonl
  for w in c2 c3 c4 c5; do python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > $F/bench_$w.log 2>&1; done
  export NRX_BENCH_OUT=$F/bench_lines_zipf.jsonl
  for z in c2 c4; do python3 bench.py --workload $z --ids zipf --steps 200 --warmup 20 --no-cpu-baseline > $F/bench_${z}_zipf.log 2>&1; done
  export NRX_BENCH_OUT=$F/bench_lines_sharded_world1.jsonl
  for w in c2 c4; do python3 bench.py --workload $w --force-sharded --shard-mode row --steps 200 --warmup 20 --no-cpu-baseline > $F/sharded_$w.log 2>&1; done
  NRX_SHARD_ONE_SIDED=1 python3 bench.py --workload c5 --force-sharded --shard-mode row --steps 200 --warmup 20 --no-cpu-baseline > $F/sharded_c5.log 2>&1
  unset NRX_BENCH_OUT
fi
