#!/usr/bin/env python3
"""profiles/traffic.json[key] <- fabric bytes per step from a tools/collect.sh summary (the `traffic` field of bench.py's roofline objects).

    update_traffic.py <key> <summary.txt> <steps> [note] [kernel-substring,kernel-substring,...]

Per kernel of this library in the summary: EA read bytes = 128 x TCC_EA0_RDREQ_128B + 64 x RDREQ_64B + 32 x the rest, EA write bytes =
64 x WRREQ_64B + 32 x the rest (separate rocprofv3 --pmc passes; requests the L2s send to the fabric, Infinity-Cache hits included --
guides/MI355X_MICROARCH.md, HBM section), times its launches per step (calls / steps).  <steps> = how many steps the profiled command ran."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import from_summary, short      # noqa: E402

key, path, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
note = sys.argv[4] if len(sys.argv) > 4 else ""
only = [x for x in (sys.argv[5].split(",") if len(sys.argv) > 5 else []) if x]      # restrict to the step's own kernels
times, ctr = from_summary(path)
rows, rd_t, wr_t, us_t = [], 0.0, 0.0, 0.0
for k, (ns, calls) in times.items():
    c = ctr.get(k, {})
    if only and not any(o in k for o in only):
        continue
    if "TCC_EA0_RDREQ_sum" not in c or "TCC_EA0_WRREQ_sum" not in c:
        continue
    r128, r64 = c.get("TCC_EA0_RDREQ_128B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0)
    rd = 128 * r128 + 64 * r64 + 32 * max(0, c["TCC_EA0_RDREQ_sum"] - r128 - r64)
    w64 = c.get("TCC_EA0_WRREQ_64B_sum", 0)
    wr = 64 * w64 + 32 * max(0, c["TCC_EA0_WRREQ_sum"] - w64)
    per = calls / steps
    if only and abs(per - round(per)) < 0.15 and round(per) >= 1:
        per = float(round(per))          # a bound call may be launched a few extra times while it is set up
    rows.append({"kernel": short(k), "launches_per_step": round(per, 3), "us": round(ns / 1e3, 1), "read_bytes": int(rd), "write_bytes": int(wr)})
    rd_t += rd * per
    wr_t += wr * per
    us_t += ns / 1e3 * per
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tj = os.path.join(root, "profiles", "traffic.json")
data = json.load(open(tj)) if os.path.exists(tj) else {}
data[key] = {"dram_read_bytes_per_launch": int(rd_t), "dram_write_bytes_per_launch": int(wr_t), "hbm_bytes_per_launch": int(rd_t + wr_t),
             "kernel_us_per_step": round(us_t, 1), "kernels": sorted(rows, key=lambda r: -r["us"] * r["launches_per_step"]),
             "source": f"{os.path.relpath(path, root)}: separate rocprofv3 --pmc passes of the same command ({steps} steps); fabric (EA) requests of every "
                       "kernel of the step x its launches per step" + (("; " + note) if note else "")}
json.dump(data, open(tj, "w"), indent=1)
print(key, json.dumps({k: v for k, v in data[key].items() if k != "kernels"}))
