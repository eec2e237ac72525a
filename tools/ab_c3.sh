#!/bin/bash
# Dev (GPU box): the C3 fused gather + cross launch under the variants of nrx_embed_dcn_v1_fwd's grouped kernel (NRX_EDCN_VARIANT / NRX_EDCN_BPC),
# alternated twice; prints ms_per_step of `bench.py --workload c3 --headline-only`.
for rep in 1 2; do
for spec in "v0=NRX_EDCN_VARIANT=0" "v1=NRX_EDCN_VARIANT=1" "v2=NRX_EDCN_VARIANT=2" "v2b3=NRX_EDCN_VARIANT=2 NRX_EDCN_BPC=3" "v2b4=NRX_EDCN_VARIANT=2 NRX_EDCN_BPC=4" "v2b8=NRX_EDCN_VARIANT=2 NRX_EDCN_BPC=8" "v2b16=NRX_EDCN_VARIANT=2 NRX_EDCN_BPC=16"; do
  label=${spec%%=*}; envs=${spec#*=}
  r=$(env $envs python3 bench.py --workload c3 --steps 300 --warmup 50 --headline-only --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step']*1e3, d['roofline'].get('kernel_us'))")
  echo "$label c3 us/step: $r"
done; done
