#!/bin/bash
# PMC + kernel-trace passes over tools/run_dcn2.py (run on the GPU box via gpurun).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/dcn2; rm -rf $F; mkdir -p $F
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"; do
  i=$((i+1)); rocprofv3 --pmc $C --kernel-trace --output-format csv -d $F/pmc$i -- python3 tools/run_dcn2.py 320 600 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $F/stats -- python3 tools/run_dcn2.py 320 600 > /dev/null 2>&1
python3 tools/summarize_profile.py $F > $F/summary.txt 2>&1
grep "dcn_v2" $F/summary.txt | awk -F'|' '{print $(NF-2), $NF}'
