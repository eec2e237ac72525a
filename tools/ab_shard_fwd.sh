#!/bin/bash
# Dev: the sharded FORWARD at world 1, same box: the bound step's engine ("feat": nrx_route_feat + nrx_gather_place_feat) against round 5's
# (NRX_SHARD_ENGINE=legacy, one-sided: nrx_route_ids_pos + nrx_gather_inbox_place).
cd "$GRAFT_REPO_ROOT" || exit 2
for rep in 1 2; do for w in ${@:-c5 c2 c3}; do
  echo "feat:   $(python3 tools/profile_sharded_step.py $w 200 fwd 2>&1 | tail -1)"
  echo "legacy: $(NRX_SHARD_ENGINE=legacy NRX_SHARD_ONE_SIDED=1 python3 tools/profile_sharded_step.py $w 200 fwd 2>&1 | tail -1)"
done; done
