#!/bin/bash
# rocprofv3 kernel stats of the DCN-v2 layer forward + backward under settled clocks (300 pairs), D = 320 and 112
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for d in 320 112; do
  rm -rf /tmp/prof_$d
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$d -- python3 $R/tools/profile_dcn2_bwd.py $d 300 > /dev/null 2>&1
  echo "== D=$d"
  python3 - <<PY
import csv, glob
f = glob.glob('/tmp/prof_$d/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>5s}  {r['Name'][:90]}")
PY
done
