cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fetch; rm -rf $O; mkdir -p $O
for m in default uncached finegrained; do
  ./tools/bin/fetch_probe $m
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d $O/$m -- ./tools/bin/fetch_probe $m > /dev/null 2>&1
  python3 - $O/$m $m <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gather_only" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(sys.argv[2], k, {c: round(sum(v[5:]) / max(1, len(v[5:]))) for c, v in d.items()})
PY
done
