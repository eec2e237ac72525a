#!/bin/bash
# Dev: same-box A/B of library variants (news_recsys_amd/lib/variants/libnrx_<name>.so; "default" = the shipped library).
#   tools/ab_variants.sh "<v1> <v2> ..." <program> [args...]
VS=$1; shift
for rep in 1 2; do for v in $VS; do
  if [ $v = default ]; then unset NRX_LIB; else export NRX_LIB=$GRAFT_REPO_ROOT/news_recsys_amd/lib/variants/libnrx_$v.so; fi
  echo "$v: $("$@" 2>&1 | grep "us " | tail -1)"
done; done
