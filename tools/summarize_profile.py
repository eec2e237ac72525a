#!/usr/bin/env python3
"""Condenses a tools/collect.sh directory into a text summary (committed under profiles/).

    summarize_profile.py <dir>             # reads <dir>/stats/**/_kernel_stats.csv and <dir>/pmc*/**/_counter_collection.csv
    summarize_profile.py --table <summary> # re-derives the per-kernel table from an existing summary.txt

Per kernel (this library's kernels only): mean duration from the --kernel-trace --stats pass, and from the separate --pmc passes
the memory-side traffic  EA read = 128 B x TCC_EA0_RDREQ_128B + 64 B x RDREQ_64B + 32 B x the rest,  EA write = 64 B x WRREQ_64B
+ 32 B x the rest  (requests the L2s send to the fabric: Infinity-Cache hits are included -- guides/MI355X_MICROARCH.md, HBM
section; FETCH_SIZE on gfx950 tallies a 128-byte request as 64 bytes and is printed only for reference), the L2 request count
and hit rate, the L1->L2 request rate, and the wave-time split."""
import collections
import csv
import glob
import re
import sys


def ours(k):
    return "at::native" not in k and "rocprim" not in k and "Cijk" not in k


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    return k.split("(")[0][:64]


def table(times, ctr):
    out = ["", "## per-kernel table (mean per launch)",
           "kernel | us | EA rd MB | EA wr MB | TB/s (rd+wr)/t | L2 req M | L2 hit % | L1->L2 rd+wr req M | req/ns | VMEM rd/wr inst K | VALU inst M | wait_any % | wait_inst % | LDS conflict % | clock GHz"]
    for k in sorted(times, key=lambda k: -times[k][0] * times[k][1]):
        c = ctr.get(k, {})
        us = times[k][0] / 1e3
        if us < 1.0:
            continue
        g = lambda n: c.get(n)
        rd = wr = None
        if g("TCC_EA0_RDREQ_sum") is not None:
            r128, r64 = g("TCC_EA0_RDREQ_128B_sum") or 0, g("TCC_EA0_RDREQ_64B_sum") or 0
            rd = (128 * r128 + 64 * r64 + 32 * max(0, g("TCC_EA0_RDREQ_sum") - r128 - r64)) / 1e6
        if g("TCC_EA0_WRREQ_sum") is not None and g("TCC_EA0_WRREQ_64B_sum") is not None:
            w64 = g("TCC_EA0_WRREQ_64B_sum")
            wr = (64 * w64 + 32 * max(0, g("TCC_EA0_WRREQ_sum") - w64)) / 1e6
        f = lambda v, p=1: "-" if v is None else f"{v:.{p}f}"
        tbs = None if rd is None or wr is None else (rd + wr) / us          # MB / us == TB/s
        req = None if g("TCC_REQ_sum") is None else g("TCC_REQ_sum") / 1e6
        hit = None if not g("TCC_REQ_sum") else 100.0 * (g("TCC_HIT_sum") or 0) / ((g("TCC_HIT_sum") or 0) + (g("TCC_MISS_sum") or 0) or 1)
        l1 = None if g("TCP_TCC_READ_REQ_sum") is None else (g("TCP_TCC_READ_REQ_sum") + (g("TCP_TCC_WRITE_REQ_sum") or 0)) / 1e6
        wc = g("SQ_WAVE_CYCLES")
        clk = None if g("GRBM_GUI_ACTIVE") is None else g("GRBM_GUI_ACTIVE") / 8 / (us * 1e3)      # the counter sums the 8 XCDs
        lds = None if not g("SQ_LDS_IDX_ACTIVE") else 100.0 * (g("SQ_LDS_BANK_CONFLICT") or 0) / g("SQ_LDS_IDX_ACTIVE")
        out.append(" | ".join([
            short(k), f"{us:.1f}", f(rd), f(wr), f(tbs, 2),
            f(req, 2), f(hit), f(l1, 2), f(None if l1 is None else l1 * 1e6 / (us * 1e3), 1),
            "-" if g("SQ_INSTS_VMEM_RD") is None else f'{g("SQ_INSTS_VMEM_RD") / 1e3:.0f}/{(g("SQ_INSTS_VMEM_WR") or 0) / 1e3:.0f}',
            f(None if g("SQ_INSTS_VALU") is None else g("SQ_INSTS_VALU") / 1e6, 2),
            f(None if not wc else 100.0 * (g("SQ_WAIT_ANY") or 0) / wc), f(None if not wc else 100.0 * (g("SQ_WAIT_INST_ANY") or 0) / wc),
            f(lds), f(clk, 2)]))
    out.append("(TB/s = (EA rd + EA wr) MB / us; req/ns = L1->L2 requests per nanosecond; wait_* = share of SQ_WAVE_CYCLES; "
               "clock = GRBM_GUI_ACTIVE / duration, with the profiler attached)")
    return "\n".join(out)


def from_summary(path):
    times, ctr = {}, collections.defaultdict(dict)
    for ln in open(path):
        p = [x.strip() for x in ln.rstrip("\n").split(" | ")]
        if len(p) == 6 and p[1].isdigit() and p[2].replace(".", "").isdigit():
            times[p[0][:90]] = (float(p[2]), int(p[1]))
        elif len(p) == 4 and p[3].startswith("mean="):
            ctr[p[0][:90]][p[1]] = float(p[3][5:])
    return times, ctr


if __name__ == "__main__":
    if sys.argv[1] == "--table":
        t, c = from_summary(sys.argv[2])
        print(table(t, c))
        sys.exit(0)
    d = sys.argv[1]
    print(f"# profile summary for {d}")
    try:
        print("command: " + open(f"{d}/command.txt").read().strip())
    except OSError:
        pass
    times = {}
    for f in glob.glob(f"{d}/stats/*/*_kernel_stats.csv"):
        print("\n## rocprofv3 --kernel-trace --stats (kernels >= 0.5 % of GPU time)")
        print("name | calls | avg_ns | min_ns | max_ns | pct")
        for r in csv.DictReader(open(f)):
            if ours(r["Name"]):
                times[r["Name"][:90]] = (float(r["AverageNs"]), int(r["Calls"]))
            if float(r["Percentage"]) >= 0.5:
                print(f'{r["Name"][:90]} | {r["Calls"]} | {float(r["AverageNs"]):.0f} | {r["MinNs"]} | {r["MaxNs"]} | {r["Percentage"]}')
    ctr = collections.defaultdict(dict)
    raw = ["\n## PMC passes, raw (mean per launch after dropping the first fifth of the launches)"]
    for f in sorted(glob.glob(f"{d}/pmc*/*/*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if ours(k):
                agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                v = v[len(v) // 5:]
                ctr[k][c] = sum(v) / len(v)
                raw.append(f"{k} | {c} | n={len(v)} | mean={sum(v) / len(v):.1f}")
    print(table(times, ctr))
    print("\n".join(raw))
