python -m pytest tests/test_hip_parity.py tests/test_models_gpu.py -m gpu -x -q -k "dcn" 2>&1 | tail -3
python tools/bench_ops.py dcn_v2 2>&1 | grep -v amdgpu.ids
