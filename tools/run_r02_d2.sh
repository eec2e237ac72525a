cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_hip_parity.py tests/test_models_gpu.py -m gpu -x -q -k "dcn" 2>&1 | tail -3
python tools/bench_ops.py dcn_v2 2>&1 | grep -v amdgpu.ids | grep bwd
O=gpurun_out/r02_d2; rm -rf $O; mkdir -p $O
for D in 320 112; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d$D -- python3 tools/profile_dcn2_bwd.py $D > $O/d$D.log 2>&1
f=$(find $O/d$D -name "*kernel_stats.csv" | head -1)
echo "== D=$D"
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "distribution" in n or float(r["AverageNs"]) < 6000: continue
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>4}  {n[:110]}')
PY
done
