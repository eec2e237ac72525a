python -m pytest tests/test_hip_parity.py tests/test_routing_properties.py tests/test_sharding_gpu.py tests/test_sharding_multirank_one_gpu.py -m gpu -x -q 2>&1 | tail -12
