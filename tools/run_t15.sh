#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_csr_bags.py tests/test_hip_parity.py tests/test_models_gpu.py tests/test_fused_sparse_adam_gpu.py tests/test_graph_capture_gpu.py -x -q -m gpu 2>&1 | tail -3
for w in "c2 uniform" "c2 zipf" "c4 uniform" "c4 zipf" "c5 uniform"; do
  set -- $w
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t15_$1$2 -- python3 tools/profile_fwd_bwd.py $1 30 $2 > gpurun_out/t15_$1$2.log 2>&1
  grep "fwd+bwd" gpurun_out/t15_$1$2.log
  python3 - gpurun_out/t15_$1$2 <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 40000 and "distribution" not in r["Name"]: print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>5}  {r["Name"][:100]}')
PY
done
