cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_fb; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_fused_sparse_adam_gpu.py tests/test_models_gpu.py tests/test_topk_retrieval.py -m gpu -x -q 2>&1 | tail -8
for w in c2 c4 c5; do
  rm -rf $O/$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 tools/profile_fwd_bwd.py $w 30 > $O/$w.log 2>&1
  f=$(find $O/$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; grep "fwd+bwd" $O/$w.log
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "distribution" in n or "FillFunctor" in n: continue
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>4}  {n[:100]}')
PY
done
