#!/bin/bash
# Dev (GPU box): the C3 fused gather + cross launch under the variants of nrx_embed_dcn_v1_fwd's grouped kernel (NRX_EDCN_VARIANT / NRX_EDCN_BPC),
# alternated twice; prints ms_per_step of `bench.py --workload c3 --headline-only`.  Specs: "<label>=<ENV=.. ENV=..>" arguments (default set below).
if [ $# -eq 0 ]; then set -- "v0=NRX_EDCN_VARIANT=0" "v1b16=NRX_EDCN_VARIANT=1" "v2b16=NRX_EDCN_VARIANT=2" "v2b12=NRX_EDCN_VARIANT=2 NRX_EDCN_BPC=12" "v4b16=NRX_EDCN_VARIANT=4" "v5b16=NRX_EDCN_VARIANT=5" "v4b8=NRX_EDCN_VARIANT=4 NRX_EDCN_BPC=8"; fi
for rep in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  r=$(env $envs python3 bench.py --workload ${WL:-c3} --steps 300 --warmup 50 --headline-only --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step']*1e3, 2))")
  echo "$label ${WL:-c3} us/step: $r"
done; done
