# measurement: variants of the placement pass of the row-sparse backward (NRX_PLACE_DEBUG: 1 no stores, 2 no loads, 4 non-temporal loads)
for w in c5 c2; do
for v in "NRX_PLACE_DEBUG=0" "NRX_PLACE_DEBUG=4" "NRX_PLACE_U=4" "NRX_PLACE_U=4 NRX_PLACE_DEBUG=4" "NRX_SPARSE_PLACE=0"; do
  echo "== $w $v"; env $v NO_PLAN_AHEAD=1 python tools/profile_fwd_bwd.py $w 100 uniform 2>&1 | grep fwd+bwd
done; done
