#!/bin/bash
# The ONE profile collector (run on the GPU box through gpurun):
#   tools/collect.sh <tag> [-s] <program> [args...]
# writes gpurun_out/<tag>/: a `rocprofv3 --kernel-trace --stats` pass and separate `--pmc` passes of the SAME command (PMC
# passes carry --kernel-trace only: never combined with another trace domain), then summary.txt (tools/summarize_profile.py).
# -s = stats pass only.  The program goes after `--` as itself (python3 <script>), never through env / bash -c.
set -u
[ -n "${GRAFT_REPO_ROOT:-}" ] || { echo "collect.sh: GRAFT_REPO_ROOT is not set (run through gpurun)"; exit 2; }
TAG=$1; shift
STATS_ONLY=0
if [ "$1" = "-s" ]; then STATS_ONLY=1; shift; fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 2
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
echo "$*" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- "$@" > $OUT/stats.log 2>&1
if [ $STATS_ONLY = 0 ]; then
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -- "$@" > $OUT/pmc$i.log 2>&1
  done
fi
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
# the raw per-pass traces (tens of MB per collection) stay on the box: gpurun copies back at most 64 MiB of gpurun_out/
[ "${KEEP_RAW:-0}" = 1 ] || rm -rf $OUT/stats $OUT/pmc[0-9]*
grep -h "fwd+bwd\|us per" $OUT/stats.log | head -5
head -60 $OUT/summary.txt
