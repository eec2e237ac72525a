#!/bin/bash
# Round-3 final artifacts (run through gpurun): whole GPU suite, smoke, the default bench command as the driver runs it, bench lines of
# every workload (+ Zipf ids), counters for every workload's headline launch and for the forward + backward legs.
cd "$GRAFT_REPO_ROOT" || exit 2
F=gpurun_out/r03_final; rm -rf $F; mkdir -p $F
python -m pytest tests -x -q -m gpu > $F/pytest_gpu.log 2>&1; tail -2 $F/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export NRX_BENCH_OUT=$F/bench_lines.jsonl
SECONDS=0; python3 bench.py > $F/bench_c2.log 2>&1; echo "default bench.py run: ${SECONDS} s"
for w in c3 c4 c5; do python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_$w.log 2>&1; done
unset NRX_BENCH_OUT
export NRX_BENCH_OUT=$F/bench_lines_zipf.jsonl
for z in c2 c4; do python3 bench.py --workload $z --ids zipf --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_${z}_zipf.log 2>&1; done
unset NRX_BENCH_OUT
tools/collect_r03.sh fb_c2 fb_c4 fb_c5 > /dev/null 2>&1
