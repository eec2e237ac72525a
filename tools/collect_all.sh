#!/bin/bash
# Round artifacts, run on the GPU box via gpurun.  Writes under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/final; rm -rf $F; mkdir -p $F
export NRX_BENCH_OUT=$F/bench_lines.jsonl
python3 bench.py > $F/bench_c2.log 2>&1
for w in c3 c4 c5; do python3 bench.py --workload $w --steps 100 --warmup 10 > $F/bench_$w.log 2>&1; done
unset NRX_BENCH_OUT
tools/collect_profile.sh final/c2 --workload c2 > /dev/null 2>&1
for w in c3 c4; do rocprofv3 --kernel-trace --stats --output-format csv -d $F/stats_$w -- python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1; done
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"; do
  i=$((i+1)); rocprofv3 --pmc $C --kernel-trace --output-format csv -d $F/dcn2/pmc$i -- python3 tools/run_dcn2.py > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $F/dcn2/stats -- python3 tools/run_dcn2.py > /dev/null 2>&1
python3 tools/summarize_profile.py $F/dcn2 > $F/dcn2_summary.txt 2>&1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rocprofv3 --pmc $C --kernel-trace --output-format csv -d $F/topk/pmc$i -- python3 tools/run_topk.py > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $F/topk/stats -- python3 tools/run_topk.py > /dev/null 2>&1
python3 tools/summarize_profile.py $F/topk > $F/topk_summary.txt 2>&1
./tools/bin/mfma_f32_probe > $F/mfma_f32_semantics.txt 2>&1
./tools/bin/mfma_peak_probe > $F/mfma_f32_valu_overlap_probe.txt 2>&1
python3 tools/bench_ops.py > $F/bench_ops.log 2>&1
python3 tools/bench_loader.py 300000 > $F/bench_loader.log 2>&1
python3 tools/bench_host_overhead.py > $F/host_overhead.log 2>&1
for z in c2 c4; do python3 bench.py --workload $z --ids zipf --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_${z}_zipf.log 2>&1; done
python3 tools/probe_bag.py > $F/probe_bag.log 2>&1
python3 tools/probe_outbuf.py > $F/probe_outbuf.log 2>&1
python3 bench.py --force-sharded --shard-mode row --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_c2_sharded_world1.log 2>&1
for w in c3 c4; do python3 - <<PY > $F/stats_$w.txt
import csv,glob
for f in glob.glob("$F/stats_$w/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"])>=1.0: print(r["Name"][:110], "|", r["Calls"], "|", round(float(r["AverageNs"])), "ns |", r["Percentage"], "%")
PY
done
ls $F
