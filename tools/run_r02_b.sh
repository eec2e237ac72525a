mkdir -p gpurun_out/r02_b
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02_b/pytest.log
tail -5 gpurun_out/r02_b/pytest.log
python bench.py --steps 100 --warmup 20 > gpurun_out/r02_b/bench_c2.log 2>&1
python bench.py --steps 100 --warmup 20 --workload c4 --no-cpu-baseline > gpurun_out/r02_b/bench_c4.log 2>&1
python bench.py --steps 50 --warmup 10 --workload c5 --no-cpu-baseline > gpurun_out/r02_b/bench_c5.log 2>&1
python bench.py --steps 50 --warmup 10 --workload c3 --no-cpu-baseline > gpurun_out/r02_b/bench_c3.log 2>&1
for f in c2 c4 c5 c3; do tail -1 gpurun_out/r02_b/bench_$f.log | cut -c1-3000; done
