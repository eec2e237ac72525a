#!/bin/bash
# Dev (GPU box): the placement pass with one feature per ring step against two (NRX_PLACE_PAIR=1: the two 64-byte halves of an upstream line
# requested back to back): kernel time (rocprofv3 --kernel-trace --stats) and fetched bytes (--pmc FETCH_SIZE, its own pass) on tools/profile_fwd_bwd.py.
cd /tmp && export TMPDIR=/tmp
for P in ${PAIRS:-0 1 0 1}; do
  export NRX_PLACE_PAIR=$P
  rm -rf /tmp/pp_s /tmp/pp_f
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_s -- python3 $GRAFT_REPO_ROOT/tools/profile_fwd_bwd.py ${WL:-c2} > /tmp/pp_s.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pp_f -- python3 $GRAFT_REPO_ROOT/tools/profile_fwd_bwd.py ${WL:-c2} > /tmp/pp_f.log 2>&1
  python3 - "$P" <<'PY'
import csv, glob, sys
P = sys.argv[1]
f = glob.glob("/tmp/pp_s/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])) if f else []:
    if "embed_bwd_place" in r["Name"]:
        print(f"PAIR={P} {r['Name'][:60]} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.2f} us")
g = glob.glob("/tmp/pp_f/**/*counter_collection.csv", recursive=True)
if g:
    tot, n = 0.0, 0
    for r in csv.DictReader(open(g[0])):
        if "embed_bwd_place" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
    if n: print(f"PAIR={P} FETCH_SIZE mean {tot / n:.0f} (units of the counter; x the guide's correction) over {n} launches")
grep = [l for l in open("/tmp/pp_s.log") if "us" in l][-3:]
print("".join(grep).strip()[:300])
PY
done
