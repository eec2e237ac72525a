#!/bin/bash
# Dev (GPU box): the current build against another build of the library (NRX_LIB), alternated runs on one box: C2 headline launch (bench.py
# --headline-only) and the bound forward + backward pass (tools/profile_fwd_bwd.py).   usage: tools/ab_lib_c2.sh <other lib> [pairs]
OTHER=$1; N=${2:-3}
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), round(d['roofline']['kernel_ms_mean']*1e3,2))"; }
for i in $(seq 1 $N); do
  echo -n "other   headline us (step, kernel): "; NRX_LIB=$OTHER timeout 200 python3 bench.py --workload ${WL:-c2} --steps 200 --warmup 20 --no-cpu-baseline --headline-only 2>/dev/null | line
  echo -n "current headline us (step, kernel): "; timeout 200 python3 bench.py --workload ${WL:-c2} --steps 200 --warmup 20 --no-cpu-baseline --headline-only 2>/dev/null | line
done
for i in $(seq 1 $N); do
  echo -n "other   "; NO_PLAN_AHEAD=1 NRX_LIB=$OTHER timeout 200 python3 tools/profile_fwd_bwd.py ${WL:-c2} 100 2>/dev/null | tail -1
  echo -n "current "; NO_PLAN_AHEAD=1 timeout 200 python3 tools/profile_fwd_bwd.py ${WL:-c2} 100 2>/dev/null | tail -1
done
