#!/bin/bash
# Copies the round artifacts that tools/collect_all.sh left under gpurun_out/final/ into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final; R=${1:-r01}
cp $F/bench_lines.jsonl profiles/${R}_bench_lines_c2_c3_c4_c5.jsonl
cp $F/c2/summary.txt profiles/${R}_c2_rocprof_summary.txt
cp $F/dcn2_summary.txt profiles/${R}_dcn_v2_mfma_summary.txt
cp $F/topk_summary.txt profiles/${R}_topk_mfma_summary.txt
grep -v amdgpu.ids $F/bench_ops.log > profiles/${R}_per_op_timings.txt
grep -v amdgpu.ids $F/bench_loader.log > profiles/${R}_loader_throughput.txt
grep -v amdgpu.ids $F/host_overhead.log > profiles/${R}_host_overhead.txt
grep -v amdgpu.ids $F/probe_bag.log > profiles/${R}_bag_probe.txt
grep -v amdgpu.ids $F/probe_outbuf.log > profiles/${R}_output_buffer_probe.txt
grep TFLOP $F/mfma_f32_valu_overlap_probe.txt > profiles/${R}_mfma_f32_valu_overlap_probe.txt
grep spread $F/mfma_f32_semantics.txt > profiles/${R}_mfma_f32_semantics.txt
cp $F/stats_c3.txt profiles/${R}_c3_kernel_stats.txt
cp $F/stats_c4.txt profiles/${R}_c4_kernel_stats.txt
for z in c2 c4; do grep '^{' $F/bench_${z}_zipf.log >> profiles/${R}_bench_lines_zipf.jsonl.tmp || true; done
mv profiles/${R}_bench_lines_zipf.jsonl.tmp profiles/${R}_bench_lines_zipf.jsonl 2>/dev/null || true
grep '^{' $F/bench_c2_sharded_world1.log > profiles/${R}_bench_c2_sharded_world1.jsonl || true
ls -la profiles/
