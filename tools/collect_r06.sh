#!/bin/bash
# Round-6 judged artifacts (run through gpurun): counters for the headline launch and for the forward + backward legs (uniform ids, full PMC
# passes; Zipf ids, stats), per-kernel stats of the sharded engine's bound training step at world 1, bench lines of every workload with
# >= 100 timed steps (+ Zipf ids), the sharded engine at world 1 with its bound fwd_bwd leg.   tools/collect_r06.sh [profiles|lines|sharded|all]
cd "$GRAFT_REPO_ROOT" || exit 2
what=${1:-all}
if [ $what = profiles ] || [ $what = all ]; then
  tools/collect.sh r06_c2 python3 bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline --headline-only > /dev/null 2>&1
  for w in c2 c4 c5; do
    NO_PLAN_AHEAD=1 tools/collect.sh r06_fb_$w python3 tools/profile_fwd_bwd.py $w 30 uniform > /dev/null 2>&1
  done
  for w in c2 c4; do
    NO_PLAN_AHEAD=1 tools/collect.sh r06_fbz_$w -s python3 tools/profile_fwd_bwd.py $w 30 zipf > /dev/null 2>&1
  done
fi
if [ $what = sharded ] || [ $what = all ]; then
  for w in c2 c4 c5; do
    tools/collect.sh r06_shst_$w python3 tools/profile_sharded_step.py $w 30 step > /dev/null 2>&1
  done
fi
if [ $what = lines ] || [ $what = all ]; then
  F=gpurun_out/r06_lines; rm -rf $F; mkdir -p $F
  export NRX_BENCH_OUT=$F/bench_lines.jsonl
  SECONDS=0; python3 bench.py > $F/bench_c2_default.log 2>&1; echo "default bench.py run: ${SECONDS} s"
  : > $F/bench_lines.jsonl
  for w in c2 c3 c4 c5; do python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > $F/bench_$w.log 2>&1; done
  export NRX_BENCH_OUT=$F/bench_lines_zipf.jsonl
  for z in c2 c4; do python3 bench.py --workload $z --ids zipf --steps 200 --warmup 20 --no-cpu-baseline > $F/bench_${z}_zipf.log 2>&1; done
  export NRX_BENCH_OUT=$F/bench_lines_sharded_world1.jsonl
  for w in c2 c4 c5; do python3 bench.py --workload $w --force-sharded --shard-mode row --steps 200 --warmup 20 --no-cpu-baseline > $F/sharded_$w.log 2>&1; done
  unset NRX_BENCH_OUT
fi
