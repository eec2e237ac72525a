#!/bin/bash
# gpurun_out/r06_* (tools/collect_r06.sh) -> profiles/r06_* (tracked): rocprof summaries (+ traffic.json for the full PMC collections) and bench lines.
set -u
cd "$(dirname "$0")/.."
[ -d gpurun_out/r06_c2 ] && tools/publish_profiles.sh r06_c2 r06_c2 c2 174 "bench.py --workload c2 --headline-only --steps 100 (+ warm-up and spread steps)" embed_fwd
for w in c2 c4 c5; do
  [ -d gpurun_out/r06_fb_$w ] && tools/publish_profiles.sh r06_fb_$w r06_fwd_bwd_$w ${w}_fwd_bwd 60 "tools/profile_fwd_bwd.py $w 30, planning inline"
  [ -d gpurun_out/r06_shst_$w ] && tools/publish_profiles.sh r06_shst_$w r06_sharded_step_$w ${w}_sharded_step_world1 36 "tools/profile_sharded_step.py $w 30 step: the bound sharded training step at world 1, row layout"
done
for w in c2 c4; do
  if [ -d gpurun_out/r06_fbz_$w ]; then
    grep -v "at::native" gpurun_out/r06_fbz_$w/summary.txt | sed "s#gpurun_out/r06_fbz_$w#(gpurun_out/r06_fbz_$w on the GPU box)#" > profiles/r06_fwd_bwd_${w}_zipf_rocprof_summary.txt
    grep -h "fwd+bwd" gpurun_out/r06_fbz_$w/stats.log | sed 's/^/# stdout of the profiled command: /' >> profiles/r06_fwd_bwd_${w}_zipf_rocprof_summary.txt
  fi
done
L=gpurun_out/r06_lines
[ -f $L/bench_lines.jsonl ] && cp $L/bench_lines.jsonl profiles/r06_bench_lines_c2_c3_c4_c5.jsonl
[ -f $L/bench_lines_zipf.jsonl ] && cp $L/bench_lines_zipf.jsonl profiles/r06_bench_lines_zipf.jsonl
[ -f $L/bench_lines_sharded_world1.jsonl ] && cp $L/bench_lines_sharded_world1.jsonl profiles/r06_bench_lines_sharded_world1.jsonl
[ -f $L/bench_c2_default.log ] && grep '^{"metric"' $L/bench_c2_default.log | tail -1 > profiles/r06_bench_line_default.json
ls -la profiles/r06_*
