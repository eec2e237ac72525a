export NRX_BENCH_HOST_STAGED=1
for w in c2 c4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --workload $w 2>&1 | grep -E '^\{|Error|error|Traceback' | cut -c1-2500
done
