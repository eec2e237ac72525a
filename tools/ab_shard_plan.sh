#!/bin/bash
# Dev: where should the owner-side plan of the sharded step run?  inline / next to the pack launch / at forward time.  World 1, bound step.
cd "$GRAFT_REPO_ROOT" || exit 2
for rep in 1 2; do for w in ${@:-c2 c5 c3}; do for m in inline backward forward; do
  echo "$m: $(NRX_SHARD_PLAN=$m python3 tools/profile_sharded_step.py $w 100 step 2>&1 | tail -1)"
done; done; done
