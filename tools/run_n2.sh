#!/bin/bash
# control-flow check of bench.py --gpus 2 on the one-GPU box: two ranks share GPU 0, collectives over gloo through host-staged buffers
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export NRX_BENCH_HOST_STAGED=1 NRX_BENCH_C5_SMALL=1
for w in c2 c4 c5; do
  echo "== $w"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 5 --warmup 2 --workload $w 2>&1 | grep -E '^\{"metric"|Error|error|Traceback|watchdog' | cut -c1-700
done
