#!/usr/bin/env python3
"""Condenses a directory of tools/collect.sh runs of tools/c2_ring_sweep.hip (one per variant) into the per-variant table committed as profiles/r02_c2_ring_sweep.md:
VGPRs / occupancy (from the kernel trace), harness time, rocprof mean time, SQ wait counters, TCP stall, DRAM bytes."""
import collections, csv, glob, re, sys

d = sys.argv[1]


def short(name):
    m = re.search(r"embed_fwd_(ring|uniform)<([^>]*)>", name.replace("(", ">(", 1) if "<" in name and ">" not in name else name)
    if not m:
        return None
    a = [x.strip() for x in m.group(2).split(",")]
    b = lambda x: "1" if x == "true" else "0"
    if m.group(1) == "ring":      # QLOG2, R, FM, STORE, NT, MINW
        return f"ring  D={4 << int(a[0])} R={a[1]} FM={b(a[2])} ST={b(a[3])} NT={b(a[4])}"
    return f"burst D={4 << int(a[0])} U={a[1]} FM={b(a[3])} ST={b(a[4])} NT={b(a[5])} (round 1)"


rows = collections.OrderedDict()
for f in glob.glob(f"{d}/stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k:
            rows.setdefault(k, {})["avg_us"] = float(r["AverageNs"]) / 1e3
            rows[k]["calls"] = int(r["Calls"])
for f in glob.glob(f"{d}/stats/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k and "vgpr" not in rows.setdefault(k, {}):
            rows[k]["vgpr"] = int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)
            rows[k]["sgpr"] = r.get("SGPR_Count", "")
            rows[k]["lds"] = r.get("LDS_Block_Size", "")
for f in sorted(glob.glob(f"{d}/pmc*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            v = v[len(v) // 5:]
            rows.setdefault(k, {})[c] = sum(v) / len(v)


# compiler resource usage (hipcc -Rpass-analysis=kernel-resource-usage, captured at build time in tools/c2_ring_sweep.resources.txt):
# the kernel-trace CSV's VGPR_Count is the arch-VGPR half of the unified file on gfx950, not the allocation
import os
res = {}
rp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c2_ring_sweep.resources.txt")
if os.path.exists(rp):
    for line in open(rp):
        parts = [x.strip() for x in line.split("|")]
        k = short(parts[0] + "(")
        if k:
            res[k] = {p.rsplit(" ", 1)[0]: p.rsplit(" ", 1)[1] for p in parts[1:]}
for k, r in rows.items():
    if k in res:
        r["vgpr"] = int(res[k]["VGPRs"])
        r["occ"] = res[k]["waves/SIMD"]
        r["sgpr"] = res[k]["SGPRs"]
        r["spill"] = res[k]["VGPR spills"]


def occ(v):
    if not v:
        return ""
    alloc = (v + 7) // 8 * 8
    return min(8, 512 // alloc)


print(f"# C2 uniform-gather sweep ({d}): 26 tables x 1M x 16 fp32, B = 65536, uniform int64 ids, recycled output buffer")
print("# rocprofv3 per kernel: --kernel-trace --stats (avg us) and separate --pmc passes (mean per dispatch, warm-up fifth dropped)")
print("# quad-cycle counters (SQ_*) are summed over all waves; DRAM MB = *_DRAM_32B x 32 B\n")
cols = ["variant", "VGPR", "SGPR", "spills", "waves/SIMD", "rocprof avg us", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "TCP_PENDING_STALL",
        "VMEM_RD", "VMEM_WR", "VALU", "DRAM rd MB", "DRAM wr MB", "EA rd req", "EA wr 64B req"]
print("| " + " | ".join(cols) + " |")
print("|" + "---|" * len(cols))
g = lambda r, k, s=1.0, f="{:.0f}": (f.format(r[k] * s) if k in r else "")
for k, r in rows.items():
    print("| " + " | ".join([k, str(r.get("vgpr", "")), str(r.get("sgpr", "")), str(r.get("spill", "")), str(r.get("occ", occ(r.get("vgpr", 0)))), g(r, "avg_us", 1, "{:.2f}"),
                             g(r, "SQ_WAVE_CYCLES", 1e-6, "{:.1f}M"), g(r, "SQ_WAIT_ANY", 1e-6, "{:.1f}M"), g(r, "SQ_WAIT_INST_ANY", 1e-6, "{:.1f}M"),
                             g(r, "TCP_PENDING_STALL_CYCLES_sum", 1e-6, "{:.1f}M"), g(r, "SQ_INSTS_VMEM_RD"), g(r, "SQ_INSTS_VMEM_WR"), g(r, "SQ_INSTS_VALU"),
                             g(r, "TCC_EA0_RDREQ_DRAM_32B_sum", 32e-6, "{:.1f}"), g(r, "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", 32e-6, "{:.1f}"),
                             g(r, "TCC_EA0_RDREQ_sum"), g(r, "TCC_EA0_WRREQ_64B_sum")]) + " |")
for name in ("c2_recycled", "c2_distinct", "c2_rows10k", "c2_rows4k", "d32_f40", "d64_f5"):
    try:
        print(f"\n## harness timings, {name}\n```")
        print(open(f"{d}/{name}.txt").read().rstrip())
        print("```")
    except OSError:
        pass
