# A/B of the planner sort (table-segmented LSD sort vs rocPRIM onesweep): forward (training form) + row-sparse backward per workload,
# then per-kernel averages of the segmented-sort run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_sort; mkdir -p $O
for w in "c2 uniform" "c2 zipf" "c4 uniform" "c4 zipf" "c5 uniform"; do
  set -- $w
  for s in segmented rocprim; do
    echo -n "$s: "; NRX_PLAN_SORT=$s python3 tools/profile_fwd_bwd.py $1 30 $2 2>&1 | grep "fwd+bwd"
  done
done
for w in "c2 uniform" "c4 uniform"; do
  set -- $w
  rm -rf $O/$1$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1$2 -- python3 tools/profile_fwd_bwd.py $1 30 $2 > $O/$1$2.log 2>&1
  f=$(find $O/$1$2 -name "*kernel_stats.csv" | head -1)
  echo "== $w"; grep "fwd+bwd" $O/$1$2.log
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "distribution" in n or "FillFunctor" in n or float(r["AverageNs"])<3000: continue
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>4}  {n[:110]}')
PY
done
