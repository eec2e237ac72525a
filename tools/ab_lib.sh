#!/bin/bash
# Dev: same-box A/B of two builds of the library: news_recsys_amd/lib/libnrx_hip_prev.so (built from the previous commit) against the
# current one, through the NRX_LIB override.  usage: tools/ab_lib.sh <program> [args...]   (prints the program's "us" lines per build)
for rep in 1 2; do
for L in prev new; do
  if [ $L = prev ]; then export NRX_LIB=$GRAFT_REPO_ROOT/news_recsys_amd/lib/libnrx_hip_prev.so; else unset NRX_LIB; fi
  echo "== $L"
  "$@" 2>&1 | grep "us "
done; done
