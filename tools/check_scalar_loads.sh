#!/bin/bash
# Dev: a codegen hazard seen with ROCm 7.2 hipcc on gfx950 (round 4, embed_bwd_small_det_kernel): for an argument block that has BOTH a byte
# array and an 8-byte array indexed by the same uniform f, the compiler formed `kernarg + f` once and addressed the 8-byte array as
#   s_load_dwordx2 sN, s[kernarg + f], soffset = 7 f, offset
# A scalar load ignores the two low bits of its base register pair, so f % 4 != 0 fetched the wrong pointer.  This lists every scalar
# load with a register offset whose base is not a kernel-entry register pair, per kernel, for a human to look at.
# usage: tools/check_scalar_loads.sh [file.hip ...]      (default: every source of the library)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/news_recsys_amd/csrc"
FILES=${@:-*.hip}
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I"$ROOT/include" --cuda-device-only -S "$f" -o /tmp/_chk.s 2>/dev/null
  n=$(awk '/^_Z[A-Za-z0-9_]*:/{name=$1} /s_load_dword.*\], s[0-9]+( offset|$)/{ if ($3 !~ /^s\[[0-9]:[0-9]\],$/) print name, $0}' /tmp/_chk.s | tee /dev/stderr | wc -l)
  echo "$f: $n scalar loads with a register offset off a computed base"
done
