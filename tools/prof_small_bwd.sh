#!/bin/bash
# Dev (GPU box): kernel durations (rocprofv3 --kernel-trace --stats) of the one-launch deterministic small backward against the float-atomic
# scatter at the reference's batch sizes, C2 shape and DSSM tower (tools/probe_dense_bwd.py).  MODES / BATCHES / NRX_LIB select.
cd /tmp && export TMPDIR=/tmp
for B in ${BATCHES:-512 2048 4096}; do for mode in ${MODES:-auto atomic}; do
  export NRX_PROBE_B=$B NRX_DENSE_BWD=$mode
  rm -rf /tmp/ps_${B}_$mode
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_${B}_$mode -- python3 $GRAFT_REPO_ROOT/tools/probe_dense_bwd.py > /tmp/ps_${B}_$mode.log 2>&1
  f=$(find /tmp/ps_${B}_$mode -name "*kernel_stats.csv" 2>/dev/null | head -1)
  echo "== B=$B $mode ${NRX_LIB##*/}"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "embed_bwd_small_det" in r["Name"] or "embed_bwd_generic" in r["Name"]:
        print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
  else tail -3 /tmp/ps_${B}_$mode.log; fi
done; done
