#!/bin/bash
# Dev (GPU box): the placement pass of the row-sparse backward, one feature per lane group (NRX_PLACE_LINES=0) against the full-line form (=1,
# embed_bwd_place_lines_kernel) -- separate kernels, alternated runs: kernel time and all library kernels per step (rocprofv3 --kernel-trace
# --stats) and fetched bytes (--pmc FETCH_SIZE, its own pass) on tools/profile_fwd_bwd.py [WL].
cd /tmp && export TMPDIR=/tmp
for C in ${CASES:-0 1 0 1 0 1}; do
  export NRX_PLACE_LINES=$C
  rm -rf /tmp/pp_s /tmp/pp_f
  NO_PLAN_AHEAD=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_s -- python3 $GRAFT_REPO_ROOT/tools/profile_fwd_bwd.py ${WL:-c2} > /tmp/pp_s.log 2>&1
  if [ -z "$NOPMC" ]; then NO_PLAN_AHEAD=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pp_f -- python3 $GRAFT_REPO_ROOT/tools/profile_fwd_bwd.py ${WL:-c2} > /tmp/pp_f.log 2>&1; fi
  python3 - "$C" <<'PY'
import csv, glob, sys
P = sys.argv[1]
f = glob.glob("/tmp/pp_s/**/*kernel_stats.csv", recursive=True)
tot = 0.0
for r in csv.DictReader(open(f[0])) if f else []:
    if "embed_bwd_place" in r["Name"]:
        print(f"LINES={P}: {r['Name'][28:90]} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.2f} us")
    if not r["Name"].startswith("void at::") and "copy" not in r["Name"].lower() and "distribution" not in r["Name"]:
        tot += float(r["TotalDurationNs"]) / 60.0 / 1e3
print(f"LINES={P}: all library kernels per step {tot:.1f} us")
g = glob.glob("/tmp/pp_f/**/*counter_collection.csv", recursive=True)
if g:
    t, n = 0.0, 0
    for r in csv.DictReader(open(g[0])):
        if "embed_bwd_place" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            t += float(r["Counter_Value"]); n += 1
    if n: print(f"LINES={P}: FETCH_SIZE mean {t / n:.0f} over {n} launches")
PY
done
