#!/bin/bash
# Runs tools/bin/c2_ring_sweep on the C2 shape un-profiled (timings), then under rocprofv3: one --kernel-trace --stats
# pass and separate --pmc passes (never combined with other trace domains).  Output: gpurun_out/<tag>/...
# usage (via gpurun): bash tools/collect_ring_sweep.sh r02_ring
set -u
TAG=${1:-r02_ring}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
S=tools/bin/c2_ring_sweep
timeout 300 $S 26 1000000 16 > $OUT/c2_recycled.txt 2>&1
timeout 300 $S 26 1000000 16 1 > $OUT/c2_distinct.txt 2>&1
timeout 300 $S 26 10000 16 > $OUT/c2_rows10k.txt 2>&1
timeout 300 $S 26 4000 16 > $OUT/c2_rows4k.txt 2>&1
timeout 300 $S 40 1000000 32 > $OUT/d32_f40.txt 2>&1
timeout 300 $S 5 10000000 64 > $OUT/d64_f5.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $S 26 1000000 16 > $OUT/stats.log 2>&1
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_64B_sum" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -- $S 26 1000000 16 > $OUT/pmc$i.log 2>&1
done
python3 tools/summarize_ring_sweep.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
