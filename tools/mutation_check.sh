#!/bin/bash
# Mutation check of the forward kernels' tail guards (run on the GPU box: `gpurun -- bash tools/mutation_check.sh`).
# Builds two deliberately broken copies of the library in a temp dir -- (1) embed_fwd_ring without its `b >= batch` exit,
# (2) embed_fwd_generic with `live` forced true -- and runs the tail-guard parity test against each through NRX_LIB.  Both runs
# must FAIL; the unmodified library must pass.  Output: gpurun_out/mutation_check.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
LOG=$OUT/mutation_check.txt; : > "$LOG"
T=$(mktemp -d)
TEST="tests/test_hip_parity.py::test_nothing_is_written_past_the_batch"
build() {   # $1 = name, $2 = file, $3 = sed expression
    local d=$T/$1; mkdir -p "$d/news_recsys_amd"; cp -r "$ROOT/news_recsys_amd/csrc" "$d/news_recsys_amd/csrc"; cp -r "$ROOT/include" "$d/include"
    sed -i "$3" "$d/news_recsys_amd/csrc/$2"
    if diff -q "$d/news_recsys_amd/csrc/$2" "$ROOT/news_recsys_amd/csrc/$2" > /dev/null; then echo "mutation $1 did not apply" | tee -a "$LOG"; return 1; fi
    make -C "$d/news_recsys_amd/csrc" -j8 ROOT="$d" > "$d/build.log" 2>&1 || { echo "mutant $1 failed to build" | tee -a "$LOG"; tail -5 "$d/build.log" >> "$LOG"; return 1; }
    echo "$d/news_recsys_amd/lib/libnrx_hip.so"
}
cd "$ROOT"
echo "== unmodified library (must pass)" | tee -a "$LOG"
python -m pytest "$TEST" -q -x 2>&1 | tail -2 | tee -a "$LOG"
M1=$(build ring nrx_embed_ring.h 's|if (b >= a->batch) return;   // the Q lanes|if (false) return;   // MUTANT: the Q lanes|') && {
    echo "== mutant 1: embed_fwd_ring without the tail exit (must FAIL)" | tee -a "$LOG"
    NRX_LIB=$M1 python -m pytest "$TEST" -q 2>&1 | tail -2 | tee -a "$LOG"; }
M2=$(build generic nrx_embed.hip '0,/const bool live = b < a.batch;/s//const bool live = true;   \/\/ MUTANT/') && {
    echo "== mutant 2: embed_fwd_generic with live = true (must FAIL)" | tee -a "$LOG"
    NRX_LIB=$M2 python -m pytest "$TEST" -q 2>&1 | tail -2 | tee -a "$LOG"; }
# ---- the full-line placement pass (embed_bwd_place_lines_kernel): the second feature of a pair reading the FIRST one's column must be caught by the
# tests that hold the pass against the float64 restatement of autograd's index_add (tests/test_plan_lds.py: both planners, both destinations)
TEST2="tests/test_plan_lds.py::test_pairs_backward_equals_sorted_backward_and_float64"
echo "== unmodified library, placement-pass tests (must pass)" | tee -a "$LOG"
python -m pytest "$TEST2" -q -x 2>&1 | tail -2 | tee -a "$LOG"
M3=$(build lines nrx_embed.hip 's|const int col = par ? a->out_col\[f1\] : a->out_col\[f0\];|const int col = a->out_col[f0];   // MUTANT: both halves read the first feature|') && {
    echo "== mutant 3: full-line placement pass, the second feature of a pair reads the first one's column (must FAIL)" | tee -a "$LOG"
    NRX_LIB=$M3 python -m pytest "$TEST2" -q 2>&1 | tail -2 | tee -a "$LOG"; }
M4=$(build pairs nrx_embed.hip 's|const float4 t_lo = row_of(p1), t_hi = row_of(p2);|const float4 t_lo = row_of(p1), t_hi = row_of(p1);   // MUTANT: the second lookup of a pair is dropped|') && {
    echo "== mutant 4: pair records, the first lookup's row added twice (must FAIL)" | tee -a "$LOG"
    NRX_LIB=$M4 python -m pytest "$TEST2" -q 2>&1 | tail -2 | tee -a "$LOG"; }
rm -rf "$T"
