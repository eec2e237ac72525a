python -m pytest tests/test_routing_properties.py tests/test_sharding_gpu.py tests/test_sharding_multirank_one_gpu.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -8
NRX_BENCH_FORCE_DIST=0 python bench.py --force-sharded --shard-mode row --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
