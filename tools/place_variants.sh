# measurement: variants of the placement pass of the row-sparse backward that the shipped library still exposes as knobs
# (NRX_PLACE_U = fetches in flight per lane, NRX_PLACE_NT = streaming loads, NRX_SPARSE_PLACE=0 = plain sorted walk).  The other rows of
# profiles/r03_place_variants.txt (no stores / no loads, the whole-line form, the clean ring, load/store wave specialisation) were
# measured with kernels that were not kept.
for w in c5 c2; do
for v in "NRX_PLACE_U=4" "NRX_PLACE_U=8" "NRX_PLACE_U=4 NRX_PLACE_NT=0" "NRX_SPARSE_PLACE=0"; do
  echo "== $w $v"; env $v NO_PLAN_AHEAD=1 python tools/profile_fwd_bwd.py $w 100 uniform 2>&1 | grep fwd+bwd
done; done
