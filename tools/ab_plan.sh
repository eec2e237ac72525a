#!/bin/bash
# Dev (GPU box): planner and forward+backward step times under library variants / env knobs, alternated twice.
# usage: tools/ab_plan.sh "<label>=<ENV=.. ENV=..>" ...     (label "base" = no env); NRX_LIB variants via NRX_LIB=news_recsys_amd/lib/variants/libnrx_X.so
WL=${WL:-"c2 c4 c5"}
for rep in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}; [ "$envs" = "$spec" ] && envs=""
  for wl in $WL; do
    p=$(env $envs python3 tools/profile_plan.py $wl 200 2>&1 | grep -o "[0-9.]* us per call")
    f=$(env $envs NO_PLAN_AHEAD=1 NRX_BENCH_C5_SMALL=${C5SMALL:-0} python3 tools/profile_fwd_bwd.py $wl 200 2>&1 | grep -o "fwd+bwd [0-9.]* us")
    echo "$label $wl plan: $p   step: $f"
  done
done; done
