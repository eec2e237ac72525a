cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_plan; mkdir -p $O
for w in c2 c4 c5; do for s in segmented rocprim; do NRX_PLAN_SORT=$s python3 tools/profile_plan.py $w 30 uniform 2>&1 | grep plan; done; done
NRX_PLAN_SORT=segmented python3 tools/profile_plan.py c2 30 zipf 2>&1 | grep plan
for w in c2 c4; do
  rm -rf $O/$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 tools/profile_plan.py $w 30 uniform > $O/$w.log 2>&1
  f=$(find $O/$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "distribution" in n or float(r["AverageNs"])<2000: continue
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>4}  {n[:110]}')
PY
done
