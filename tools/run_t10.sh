#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_hip_parity.py tests/test_models_gpu.py -x -q -m gpu -k "dcn" 2>&1 | tail -3
python tools/bench_ops.py dcn_v2 2>&1 | grep "dcn_v2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t10 -- python3 tools/profile_dcn2_bwd.py 320 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/t10/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 20000 and "distribution" not in r["Name"]: print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>5}  {r["Name"][:100]}')
PY
