#!/bin/bash
# Dev (round 6): the Zipf forward + backward leg on (a) the round-4 tree (tools/_r4tree, built from a4bbe40; not committed), (b) HEAD (fence-free
# hand-over of the multi-item rows' partials), (c) HEAD with the separate combine launch (variants/libnrx_combsep.so).  Run through gpurun.
#   tools/zipf_ab.sh [profile]     profile: per-kernel stats of every form first
cd "$GRAFT_REPO_ROOT" || exit 2
COMB=$GRAFT_REPO_ROOT/news_recsys_amd/lib/variants/libnrx_combsep.so
if [ "${1:-}" = profile ]; then
for w in c2 c4; do
  [ -d tools/_r4tree ] && NO_PLAN_AHEAD=1 tools/collect.sh zipf_r4_$w -s python3 tools/_r4tree/tools/profile_fwd_bwd.py $w 30 zipf > /dev/null 2>&1
  NO_PLAN_AHEAD=1 tools/collect.sh zipf_head_$w -s python3 tools/profile_fwd_bwd.py $w 30 zipf > /dev/null 2>&1
  NRX_LIB=$COMB NO_PLAN_AHEAD=1 tools/collect.sh zipf_combsep_$w -s python3 tools/profile_fwd_bwd.py $w 30 zipf > /dev/null 2>&1
done
for t in r4 head combsep; do for w in c2 c4; do [ -d gpurun_out/zipf_${t}_$w ] || continue; echo "=== $t $w"; grep "fwd+bwd" gpurun_out/zipf_${t}_$w/stats.log; sed -n '/^## rocprofv3/,/^## per-kernel/p' gpurun_out/zipf_${t}_$w/summary.txt | grep -v "at::native" | cut -c1-200; done; done > gpurun_out/zipf_ab.txt 2>&1
fi
# un-profiled timings, alternating
for rep in 1 2; do
  for w in c2 c4; do
    for d in zipf uniform; do
    [ -d tools/_r4tree ] && echo "r4: $(NO_PLAN_AHEAD=1 python3 tools/_r4tree/tools/profile_fwd_bwd.py $w 100 $d 2>&1 | tail -1)"
    echo "head: $(NO_PLAN_AHEAD=1 python3 tools/profile_fwd_bwd.py $w 100 $d 2>&1 | tail -1)"
    echo "combsep: $(NRX_LIB=$COMB NO_PLAN_AHEAD=1 python3 tools/profile_fwd_bwd.py $w 100 $d 2>&1 | tail -1)"
    done
  done
done >> gpurun_out/zipf_ab.txt 2>&1
tail -24 gpurun_out/zipf_ab.txt
