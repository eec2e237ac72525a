#!/usr/bin/env python3
"""Regression gate over the committed bench lines: every timed leg of a round's `profiles/rNN_bench_lines*.jsonl` against the previous
round's file of the same name, leg by leg (a leg = a numeric entry whose key names a duration: *_ms, *_us, ms_per_*, us_per_*, .us).
Exit code 1 when any GPU-time leg is more than --tol (default 10 %) slower -- the check round 5 did not have when its Zipf legs went from
281 / 410 us to 414 / 674 us unnoticed.  Host-time legs (`*host_us*`, and the eager module-path legs, which are launched from Python step by
step) are listed but never fail the gate: they follow the box's CPU.

    tools/compare_bench_lines.py                 # newest round in profiles/ against the one before it
    tools/compare_bench_lines.py 6 5             # round 6 against round 5
    tools/compare_bench_lines.py a.jsonl b.jsonl # two files (new, old)
Lines are matched by (workload tag = the text of config.workload up to the first ':', n_gpus, config.ids when present)."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIME_KEY = re.compile(r"(^|[._])(ms|us)($|[._])|_ms$|_us$|ms_per|us_per")


def legs(d, pre=""):
    out = {}
    if isinstance(d, dict):
        for k, v in d.items():
            out.update(legs(v, f"{pre}{k}."))
    elif isinstance(d, (int, float)) and not isinstance(d, bool):
        key = pre[:-1]
        last2 = ".".join(key.split(".")[-2:])
        if TIME_KEY.search(last2) and "cpu_baseline" not in key and "stream_copy" not in key:
            out[key] = float(d)
    return out


def tag(line):
    cfg = line.get("config", {})
    wl = str(cfg.get("workload", "?")).split(":")[0].strip()
    ids = "zipf" if "zipf" in json.dumps(cfg).lower() else "uniform"
    lay = cfg.get("layout")
    sharded = "sharded" if (isinstance(lay, dict) or "row-sharded over" in str(cfg.get("workload", ""))) else "direct"
    return (wl, line.get("n_gpus", 1), ids, sharded)


def load(path):
    res = {}
    for l in open(path):
        l = l.strip()
        if l.startswith("{"):
            d = json.loads(l)
            res[tag(d)] = legs(d)
    return res


def compare(new_path, old_path, tol):
    new, old = load(new_path), load(old_path)
    bad = 0
    print(f"== {os.path.basename(new_path)}  vs  {os.path.basename(old_path)}")
    for t in sorted(new):
        if t not in old:
            print(f"  {t}: new line (no counterpart)")
            continue
        for k in sorted(new[t]):
            if k not in old[t] or old[t][k] <= 0:
                continue
            r = new[t][k] / old[t][k]
            # host-time legs: named so, or the eager module-path legs (launched from Python step by step: GPU time = host time there)
            host = "host" in k or re.search(r"module_path\..*(autograd_us|hook_us|step_us|graphed_us)", k) is not None
            flag = ""
            if r > 1.0 + tol:
                flag = "  (host time: not gated)" if host else "  <-- SLOWER"
                bad += 0 if host else 1
            elif r < 1.0 - tol:
                flag = "  faster"
            if flag:
                print(f"  {t[0]:>3} {t[2]:>7} {k:<70} {old[t][k]:>10.4f} -> {new[t][k]:>10.4f}  x{r:.2f}{flag}")
    return bad


def round_files(n):
    return sorted(glob.glob(os.path.join(ROOT, "profiles", f"r{int(n):02d}_bench_lines*.jsonl")))


def main(argv):
    tol = 0.10
    args = [a for a in argv if not a.startswith("--tol")]
    for a in argv:
        if a.startswith("--tol="):
            tol = float(a.split("=")[1])
    if len(args) == 2 and all(a.endswith(".jsonl") for a in args):
        sys.exit(1 if compare(args[0], args[1], tol) else 0)
    if len(args) == 2:
        new_r, old_r = int(args[0]), int(args[1])
    else:
        rounds = sorted({int(re.search(r"r(\d+)_bench_lines", f).group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_lines*.jsonl"))})
        if len(rounds) < 2:
            print("fewer than two rounds of bench lines in profiles/")
            sys.exit(0)
        new_r, old_r = rounds[-1], rounds[-2]
    bad = 0
    for f in round_files(new_r):
        g = os.path.join(os.path.dirname(f), os.path.basename(f).replace(f"r{new_r:02d}_", f"r{old_r:02d}_"))
        if os.path.exists(g):
            bad += compare(f, g, tol)
        else:
            print(f"== {os.path.basename(f)}: no round-{old_r} file of that name")
    print(f"{bad} GPU-time leg(s) more than {tol:.0%} slower than round {old_r}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main(sys.argv[1:])
