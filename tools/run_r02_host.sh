python tools/bench_host_overhead.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
