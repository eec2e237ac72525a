#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_csr_bags.py tests/test_hip_parity.py tests/test_models_gpu.py tests/test_fused_sparse_adam_gpu.py tests/test_graph_capture_gpu.py -x -q -m gpu > gpurun_out/t16.log 2>&1
grep -n "^E \|passed\|failed\|Error" gpurun_out/t16.log | head -30 | cut -c1-250
