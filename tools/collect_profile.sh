#!/bin/bash
# Collects the judged artifacts for one bench configuration on the GPU box (run via gpurun):
#   kernel-trace --stats summary + separate PMC passes (never combined with other trace domains).
# usage: tools/collect_profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --headline-only $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_BUSY_CYCLES" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr" \
         "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -- $B > $OUT/pmc$i.log 2>&1
done
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
