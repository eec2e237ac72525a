#!/bin/bash
# Dev (GPU box): the row-sharded engine at world = 1 (bench.py --force-sharded), buffer path against one-sided placement, per workload; appends the
# bench lines to gpurun_out/r04_sharded_world1.jsonl and prints the step times.
OUT=gpurun_out/r04_sharded_world1.jsonl; : > $OUT
for wl in c2 c4 c5; do
  for os in 0 1; do
    line=$(NRX_SHARD_ONE_SIDED=$os NRX_BENCH_C5_SMALL=${C5SMALL:-0} python3 bench.py --workload $wl --force-sharded --shard-mode ${MODE:-row} --steps 200 --warmup 20 --headline-only --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$line" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
d['one_sided_placement'] = bool($os)
print(json.dumps(d))" >> $OUT
    echo "$wl one_sided=$os: $(echo "$line" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us/step')")"
  done
done
