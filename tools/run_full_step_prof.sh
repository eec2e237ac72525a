#!/bin/bash
# per-kernel GPU time of the C2 ranker step with the MLP weight gradients on nrx_linear_wgrad (leg a' of tools/bench_full_step_c2.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export LEGS=a1
rm -rf /tmp/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -- python3 $R/tools/bench_full_step_c2.py > /tmp/prof_step.log 2>&1
tail -2 /tmp/prof_step.log
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_step/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 70.0        # 40 warm-up + 30 timed steps
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"GPU time per step {tot / steps / 1e3:.0f} us")
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step  {float(r['AverageNs'])/1e3:8.1f} us x {float(r['Calls'])/steps:5.1f}/step  {r['Name'][:110]}")
PY
