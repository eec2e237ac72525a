#!/bin/bash
# Round-3 judged artifacts (run through gpurun): counters for every bench workload's headline launch and for the forward + backward leg.
#   tools/collect_r03.sh [c2 c3 c4 c5 fb_c2 fb_c4 fb_c5 ...]
cd "$GRAFT_REPO_ROOT" || exit 2
for w in "$@"; do
  case $w in
    fb_*) wl=${w#fb_}; NO_PLAN_AHEAD=1 tools/collect.sh r03_$w python3 tools/profile_fwd_bwd.py $wl 30 uniform > /dev/null 2>&1 ;;
    *)    tools/collect.sh r03_$w python3 bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --headline-only > /dev/null 2>&1 ;;
  esac
  rm -rf gpurun_out/r03_$w/stats gpurun_out/r03_$w/pmc*/
done
