#!/bin/bash
# Dev (GPU box): batch sweep of the default-mode (dense-gradient) embedding-only step, deterministic sorted reduction against the float-atomic
# scatter (tools/probe_dense_bwd.py; NRX_DENSE_BWD forces the mode).  Prints the second (warm) pass of each case.
for B in ${BATCHES:-512 2048 8192 32768 65536}; do
  for mode in sorted atomic; do
    NRX_PROBE_B=$B NRX_DENSE_BWD=$mode timeout 120 python3 tools/probe_dense_bwd.py 2>/dev/null | tail -2 | sed "s/^/B=$B $mode | /"
  done
done
