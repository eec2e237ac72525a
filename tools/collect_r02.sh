#!/bin/bash
# Round-2 artifacts, run on the GPU box via gpurun.  Writes under gpurun_out/r02_final/ (copied into profiles/ by
# tools/publish_profiles_r02.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/r02_final; rm -rf $F; mkdir -p $F
stats() {  # stats <dir>: kernels >= 3 us of a --kernel-trace --stats run
python3 - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "distribution" in n or "FillFunctor" in n or float(r["AverageNs"]) < 3000: continue
        print(f'{float(r["AverageNs"]) / 1e3:9.1f} us x{r["Calls"]:>5}  {n[:120]}')
PY
}
export NRX_BENCH_OUT=$F/bench_lines.jsonl
python3 bench.py > $F/bench_c2.log 2>&1
for w in c3 c4 c5; do python3 bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_$w.log 2>&1; done
unset NRX_BENCH_OUT
export NRX_BENCH_OUT=$F/bench_lines_zipf.jsonl
for z in c2 c4; do python3 bench.py --workload $z --ids zipf --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_${z}_zipf.log 2>&1; done
unset NRX_BENCH_OUT
export NRX_BENCH_OUT=$F/bench_lines_sharded_world1.jsonl
for z in c2 c4; do python3 bench.py --workload $z --force-sharded --shard-mode row --steps 100 --warmup 10 --no-cpu-baseline > $F/bench_${z}_sh1.log 2>&1; done
unset NRX_BENCH_OUT
tools/collect_profile.sh r02_final/c2 --workload c2 > /dev/null 2>&1
tools/collect_profile.sh r02_final/c4 --workload c4 > /dev/null 2>&1
bash tools/collect_ring_sweep.sh r02_final/ring > /dev/null 2>&1
{
for w in "c2 uniform" "c2 zipf" "c4 uniform" "c5 uniform"; do
  set -- $w
  rocprofv3 --kernel-trace --stats --output-format csv -d $F/fb_$1$2 -- python3 tools/profile_fwd_bwd.py $1 30 $2 > $F/fb_$1$2.log 2>&1
  echo "== forward (training form) + row-sparse backward, workload $1, $2 ids (30 warm-up + 30 timed steps; per-kernel averages)"
  grep "fwd+bwd" $F/fb_$1$2.log
  stats $F/fb_$1$2
done
} > $F/fwd_bwd_kernel_stats.txt 2>&1
{
for D in 320 112; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $F/dcn2_$D -- python3 tools/profile_dcn2_bwd.py $D > $F/dcn2_$D.log 2>&1
  echo "== DCN-v2 layer forward + backward, B = 65536, D = $D (per-kernel averages)"
  stats $F/dcn2_$D
done
} > $F/dcn_v2_bwd_kernel_stats.txt 2>&1
{
for z in c2 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $F/sh1_$z -- python3 bench.py --workload $z --force-sharded --shard-mode row --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  echo "== row-sharded engine at world = 1, workload $z (per-kernel averages of one step's launches)"
  stats $F/sh1_$z
done
} > $F/sharded_world1_kernel_stats.txt 2>&1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"; do
  i=$((i+1)); rocprofv3 --pmc $C --kernel-trace --output-format csv -d $F/dcn2/pmc$i -- python3 tools/profile_dcn2_bwd.py 320 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $F/dcn2/stats -- python3 tools/profile_dcn2_bwd.py 320 > /dev/null 2>&1
python3 tools/summarize_profile.py $F/dcn2 > $F/dcn2_summary.txt 2>&1
python3 tools/bench_ops.py > $F/bench_ops.log 2>&1
python3 tools/bench_host_overhead.py > $F/host_overhead.log 2>&1
ls $F
