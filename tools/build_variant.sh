#!/bin/bash
# Dev: build a variant of the library with extra compiler flags into news_recsys_amd/lib/variants/libnrx_<name>.so (ships to the GPU box;
# select with NRX_LIB).  usage: tools/build_variant.sh <name> <flags...>     e.g. tools/build_variant.sh t1024 -DNRX_SEG_THREADS=1024
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
D=$(mktemp -d)
mkdir -p "$D/news_recsys_amd" "$ROOT/news_recsys_amd/lib/variants"
cp -r "$ROOT/news_recsys_amd/csrc" "$D/news_recsys_amd/csrc"; cp -r "$ROOT/include" "$D/include"
make -C "$D/news_recsys_amd/csrc" -j8 FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fvisibility=hidden -I$D/include -Wall -Wno-unused-function $*" > "$D/build.log" 2>&1 || { tail -20 "$D/build.log"; exit 1; }
cp "$D/news_recsys_amd/lib/libnrx_hip.so" "$ROOT/news_recsys_amd/lib/variants/libnrx_$NAME.so"
rm -rf "$D"
echo "built news_recsys_amd/lib/variants/libnrx_$NAME.so"
