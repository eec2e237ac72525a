cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_hip_parity.py tests/test_routing_properties.py tests/test_sharding_gpu.py tests/test_sharding_multirank_one_gpu.py -m gpu -x -q 2>&1 | tail -12
O=gpurun_out/r02_sh; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python3 bench.py --force-sharded --shard-mode row --workload c4 --steps 50 --warmup 10 --no-cpu-baseline > $O/c4.log 2>&1
f=$(find $O/c4 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "distribution" in n or "FillFunctor" in n: continue
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>4}  {n[:100]}')
PY
