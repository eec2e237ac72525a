#!/usr/bin/env python3
"""bench.py -- impressions/sec of the embedding hot path at batch 65536 (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: a plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) starts the N rank
processes itself (self_launch below: child processes, no exec, the launcher never touches the GPU), relays rank 0's one
JSON line and exits with the worst rank's code.

A "step" is one pass of the hot path over one batch of synthetic MIND-shaped impressions already
resident in HBM: fused multi-table gather (+pool) -> concat (+ the model's interaction epilogue).
Default workload = BASELINE.json configs[1] ("c2": DeepFM-shaped, 26 sparse x 1M rows x 16, B=65536,
uniform ids): gather -> [B,416] concat + fused FM logit.  A fresh id batch (from a pool of 8) is used
every step so no step re-reads the previous step's rows from cache.

N > 1: every rank owns B=65536 impressions (weak scaling; `value`).  The headline layout is "row" for EVERY workload: every
table row-sharded (row r on rank r % N) with RCCL all-to-all id routing + row return / owner-side pooling
(news_recsys_amd/sharding.py, shard_step.py) -- what north_star scales.  "auto" = planner (tables of at most 256 MiB replicated,
larger ones row-sharded) goes under "other_layout" with the number of tables it replicated; "strong_scaling" re-times the
headline layout with the global batch fixed at 65536; "a2a" times the row-return all-to-all alone (GB/s per rank and per xGMI
link, next to the 7 x 153 GB/s of the links); "fwd_bwd" is the bound sharded training step (forward + gradient exchange +
owner-side row-sparse reduction) with its own roofline.  The N > 1 line also carries `rccl_ranks` (the process group's own world
size and backend) and `scaling_vs_1gpu` (value / N over the DIRECT single-GPU path's committed number of the same workload).
A watchdog prints the headline line if a secondary leg stalls.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# ------------------------------------------------------------------------------------ self-launch (N > 1 without torchrun)
def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(n: int, child_cmd: list, timeout_s: float, extra_env: dict | None = None):
    """Start `n` rank processes of `child_cmd` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their
    environment, one per GPU), wait for all of them and return (worst return code, rank 0's stdout lines, note).
    Plain child processes: no exec anywhere, and the caller has not touched the GPU.  Rank 0's stdout is collected,
    everything else the ranks print goes to this process's stderr.  A rank that fails takes the others down after a
    short grace period (they would sit in a collective until the RCCL timeout); a job that outlives `timeout_s` is
    killed -- exactly the PIDs started here -- and reported with a non-zero code."""
    import subprocess
    import threading
    port = _free_port()
    procs, lines = [], []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "NRX_BENCH_CHILD": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(child_cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr, text=True, start_new_session=True))

    def pump():
        for ln in procs[0].stdout:
            lines.append(ln.rstrip("\n"))
    th = threading.Thread(target=pump, daemon=True)
    th.start()

    def kill_all():
        import signal
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)       # the child's own session (start_new_session): nothing else is in it
                except ProcessLookupError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:       # noqa: BLE001
                pass

    deadline = time.monotonic() + timeout_s
    first_fail = None
    note = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        now = time.monotonic()
        if bad and first_fail is None:
            first_fail = now
        if first_fail is not None and now - first_fail > 15.0:
            note = f"rank {bad[0][0]} exited with code {bad[0][1]}; the remaining ranks were killed"
            kill_all()
            break
        if now > deadline:
            note = f"the {n}-rank job did not finish within {timeout_s:.0f} s and was killed"
            kill_all()
            break
        time.sleep(0.05)
    th.join(timeout=5)
    rcs = [p.returncode if p.returncode is not None else -9 for p in procs]
    worst = 0
    for rc in rcs:
        if rc != 0:
            worst = rc if rc > 0 else 128 + abs(rc)
            break
    if note and worst == 0:
        worst = 124
    return worst, lines, note


def self_launch(args, argv: list) -> int:
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: be the launcher.  Job 1 runs the N ranks
    with --headline-only (the timed region of the contract and nothing else: a clean, separately exiting measurement);
    job 2 re-runs them with the secondary legs (other layout, strong scaling, all-to-all probe) and only its secondary
    fields are merged into job 1's line -- if job 2 fails or stalls it is killed, a note says so, and the exit code stays
    that of the measurement.  NRX_BENCH_LAUNCH_TIMEOUT (s, default 1500) bounds each job."""
    n = args.gpus
    timeout_s = float(os.environ.get("NRX_BENCH_LAUNCH_TIMEOUT", "1500"))
    me = [sys.executable, os.path.abspath(__file__)]
    base = [a for a in argv if a != "--headline-only"]
    rc, lines, note = run_ranks(n, me + base + ["--headline-only"], timeout_s)
    head = None
    for ln in lines:
        if ln.startswith("{"):
            try:
                head = json.loads(ln)
            except ValueError:
                continue
    if rc != 0 or head is None:
        for ln in lines:
            print(ln, flush=True)
        sys.stderr.write(f"bench.py: the {n}-rank headline job failed (rc {rc}){': ' + note if note else ''}\n")
        return rc or 1
    head["launcher"] = (f"self-launched: {n} rank processes started by `python bench.py --gpus {n}` (RANK / LOCAL_RANK / WORLD_SIZE / "
                        "MASTER_ADDR=127.0.0.1 in their environment), headline job with --headline-only, secondary legs as a second job")
    if not args.headline_only:
        rc2, lines2, note2 = run_ranks(n, me + base, min(timeout_s, args.secondary_timeout + 600.0))
        sec = None
        for ln in lines2:
            if ln.startswith("{"):
                try:
                    sec = json.loads(ln)
                except ValueError:
                    continue
        if sec is not None:
            for k in ("other_layout", "strong_scaling", "a2a", "fwd_bwd", "secondary_note"):
                if k in sec:
                    head[k] = sec[k]
        if rc2 != 0 or sec is None:
            head["secondary_note"] = (f"the secondary-legs job ended with code {rc2}" + (f" ({note2})" if note2 else "")
                                      + "; the headline above comes from its own job, which exited cleanly")
    print(json.dumps(head), flush=True)
    if os.environ.get("NRX_BENCH_OUT"):
        with open(os.environ["NRX_BENCH_OUT"], "a") as f:
            f.write(json.dumps(head) + "\n")
    return 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the secondary legs (distinct output buffers, fwd+bwd)")
    ap.add_argument("--ids", default="uniform", choices=["uniform", "zipf"],
                    help="id distribution: uniform (headline, cache-hostile) or Zipf(1.05) popularity (MIND-like); N=1 only")
    ap.add_argument("--force-sharded", action="store_true", help="run the row-sharded engine even at N=1 (testing)")
    ap.add_argument("--shard-mode", default="default", choices=["default", "row", "auto"],
                    help="N>1 headline layout: 'row' = every table row-sharded; 'auto' = planner (tables <= 256 MiB replicated, "
                         "larger ones row-sharded); 'default' = row for every workload.  The other layout is timed "
                         "too (field `other_layout`, which says how many tables it replicated).")
    ap.add_argument("--secondary-timeout", type=float, default=240.0,
                    help="N>1: seconds the secondary legs (other layout, strong scaling, a2a probe) may take before the "
                         "headline line is printed without them")
    return ap.parse_args(argv)


if __name__ == "__main__" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("NRX_BENCH_CHILD") != "1":
    # before torch is imported and before anything can touch the GPU: N > 1 asked of a plain `python bench.py` -> launch the ranks
    _a = parse_args()
    if _a.gpus > 1:
        sys.exit(self_launch(_a, sys.argv[1:]))

import numpy as np      # noqa: E402
import torch            # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 measured achievable
BATCH = 65536


# ------------------------------------------------------------------------------------ workloads
def workload_spec(name: str):
    """(features, description).  feature = dict(name, rows, dim, bag_len)."""
    if name == "c2":      # DeepFM: 26 sparse x 1M rows, D=16
        rows = int(os.environ.get("NRX_BENCH_C2_ROWS", 1_000_000))      # dev knob (cache-resident tables); headline = 1M
        feats = [dict(name=f"C{i:02d}", rows=rows, dim=16, bag=0) for i in range(26)]
        return feats, "c2: DeepFM 26 sparse x 1M rows, D=16, B=65536, uniform ids; gather->concat[B,416] + fused FM logit"
    if name == "c3":      # DCN: 5 feats D=64, news table 100M rows
        rows = dict(item_id=100_000_000, user_id=1_000_000, category=18, subcategory=270, user_click_category=18)
        feats = [dict(name=k, rows=v, dim=64, bag=0) for k, v in sorted(rows.items())]
        return feats, "c3: DCN 5 feats D=64 (item_id 100M rows), B=65536; gather->concat into [B,640] + 2-layer cross written next to x (2 launches)"
    if name == "c4":      # DSSM: user_id 10M, item_id 200k, history L=50 shares item table, D=16
        feats = [dict(name="item_id", rows=200_000, dim=16, bag=0),
                 dict(name="user_history", rows=200_000, dim=16, bag=50, share="item_id"),
                 dict(name="user_id", rows=10_000_000, dim=16, bag=0)]
        return feats, "c4: DSSM user_id 10M + history L=50 (mean-pool, shares 200k news table) + item_id, D=16, B=65536"
    if name == "c5":      # WideDeep: 40 feats 1k..500M rows D=32 (one GPU holds the <=16M-row tables only)
        rows = [int(round(1e3 * (5e5) ** (i / 39))) for i in range(40)]
        feats = [dict(name=f"W{i:02d}", rows=r, dim=32, bag=0) for i, r in enumerate(rows)]
        return feats, "c5: WideDeep 40 feats 1k..500M rows D=32"
    raise SystemExit(f"unknown workload {name}")


def algorithmic_bytes_per_impression(feats, fm: bool, cross_dim: int = 0) -> int:
    """int64 ids 8 B, fp32 rows, each looked-up row counted once, output written once.
    single-valued: 8 + 4D (row read) + 4D (write); bag (L padded, all valid): L*(8+4) + L*4D + 4D."""
    total = 0
    for f in feats:
        D, L = f["dim"], f["bag"]
        total += (8 + 4 * D + 4 * D) if L == 0 else (L * 12 + L * 4 * D + 4 * D)
    if fm:
        total += 4
    total += 4 * cross_dim          # fused cross output written next to x
    return total


def granule_bytes_per_impression(feats, fm: bool, cross_dim: int = 0) -> int:
    """The same lookups priced at the memory system's granules: a random row read costs whole 128-byte lines (every fabric read
    request of these kernels is 128 bytes: TCC_EA0_RDREQ_128B == RDREQ in profiles/), writes go out in 64-byte requests, ids and
    masks stream.  For 64-byte rows (D = 16) this is what the counters measure (C2: 341 MB per launch vs 232 MB algorithmic)."""
    total = 0
    for f in feats:
        D, L = f["dim"], f["bag"]
        row_rd = -(-4 * D // 128) * 128
        row_wr = -(-4 * D // 64) * 64
        total += (8 + row_rd + row_wr) if L == 0 else (L * 12 + L * row_rd + row_wr)
    if fm:
        total += 4
    total += 4 * cross_dim
    return total


def backward_bytes_per_impression(feats, fm: bool) -> int:
    """Algorithmic bytes of the row-sparse backward, same conventions as SURVEY 8d (ids 8 B, fp32 rows, every lookup counted once):
    per lookup the planner reads the id (8) and the reduction reads the upstream row (4D; a bag lookup reads its sample's row and
    its 4-byte weight) and -- FM fields -- the forward value (4D); per unique row one gradient row (4D) and one key (8) are
    written, counted per lookup (uniform ids: ~97 % of the lookups are distinct rows); per sample the FM field sums (4D) and
    g_fm (4) are read.  Sorting traffic is NOT counted: it is overhead of the method, not of the problem."""
    total = 0
    for f in feats:
        D, L = f["dim"], f["bag"]
        n = max(L, 1)
        total += n * (8 + 4 * D + (4 if L else 0) + (4 * D if fm else 0) + 4 * D + 8)
    if fm:
        total += 4 * feats[0]["dim"] + 4
    return total


# ------------------------------------------------------------------------------------ single-GPU runner
def draw_ids(rows: int, shape, device, gen, dist: str) -> torch.Tensor:
    """Synthetic ids in [1, rows): 'uniform' (headline: cache-hostile) or 'zipf' = Zipf(1.05) popularity
    ranks clipped to the table (MIND-like: a few hot news ids), by inverse-CDF sampling on the device."""
    if dist == "uniform":
        return torch.randint(1, rows, shape, device=device, generator=gen)
    u = torch.rand(shape, device=device, generator=gen, dtype=torch.float64)
    a = 1.05
    n = float(rows - 1)
    # continuous power-law on [1, n]: F^-1(u) = (1 + u (n^(1-a) - 1))^(1/(1-a))
    r = (1.0 + u * (n ** (1.0 - a) - 1.0)) ** (1.0 / (1.0 - a))
    return r.long().clamp_(1, rows - 1)


class SingleGpuPath:
    def __init__(self, wl: str, device, seed: int, n_pool: int = 8, id_dist: str = "uniform"):
        from news_recsys_amd import ops
        from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_SPARSE
        self.ops = ops
        self.wl = wl
        feats, self.desc = workload_spec(wl)
        if wl == "c5":
            # all 40 tables (224 GB of fp32 rows) fit one 288 GB MI355X; a smaller device keeps the tables it can hold
            need = sum(f["rows"] * f["dim"] * 4 for f in feats) + (6 << 30)
            free = torch.cuda.mem_get_info(device)[0]
            if os.environ.get("NRX_BENCH_C5_SMALL") == "1" or free < need:
                feats = [f for f in feats if f["rows"] <= 16_000_000]
                self.desc += f" -- this run keeps the {len(feats)} tables <= 16M rows ({free >> 30} GiB free, {need >> 30} GiB needed for all 40)"
            else:
                self.desc += " -- all 40 tables resident on one GPU (224 GB)"
            feats = sorted(feats, key=lambda f: -f["rows"])       # allocate the big tables first
        self.feats = feats
        gen = torch.Generator(device=device).manual_seed(seed)
        self.tables, tindex = [], {}
        for f in feats:
            tname = f.get("share", f["name"])
            if tname in tindex:
                continue
            t = torch.empty((f["rows"], f["dim"]), dtype=torch.float32, device=device)
            for r0 in range(0, f["rows"], 1 << 26):               # in slabs: no multi-GB temporaries from the generator
                t[r0:r0 + (1 << 26)].normal_(generator=gen)
            t[0].zero_()
            tindex[tname] = len(self.tables)
            self.tables.append(t)
        slots, col = [], 0
        self.fm = wl == "c2"
        for f in feats:
            kind = NRX_BAG_MASKED_MEAN if f["bag"] else NRX_SPARSE
            slots.append(ops.Slot(f["name"], kind, tindex[f.get("share", f["name"])], f["dim"], f["bag"], col,
                                  fm_field=int(self.fm)))
            col += f["dim"]
        self.width = col
        self.plan = ops.EmbedPlan(slots, out_width=col, use_fm=self.fm)
        self.cross = None
        if wl == "c3":
            self.cross_w = torch.randn(2, col, device=device, generator=gen) / col ** 0.5
            self.cross_b = torch.zeros(2, col, device=device)
            self.cross = True
        self.pool = []
        for _ in range(n_pool):
            ins, ws = [], []
            for f in feats:
                if f["bag"]:
                    x = draw_ids(f["rows"], (BATCH, f["bag"]), device, gen, id_dist)
                    w = torch.ones((BATCH, f["bag"]), dtype=torch.float32, device=device)
                    if os.environ.get("NRX_BENCH_PADDED_HISTORY") == "1":
                        # dev knob (never the headline: SURVEY 8d prices the bag as all valid): histories of uniform length 0 .. L padded with id 0
                        # and mask 0, as the reference's DataReader leaves them
                        n_valid = torch.randint(0, f["bag"] + 1, (BATCH, 1), device=device, generator=gen)
                        w = (torch.arange(f["bag"], device=device)[None, :] < n_valid).float()
                        x = x * w.long()
                    ins.append(x)
                    ws.append(w)
                else:
                    ins.append(draw_ids(f["rows"], (BATCH,), device, gen, id_dist))
                    ws.append(None)
            self.pool.append((ins, ws))
        self.bytes_per_impr = algorithmic_bytes_per_impression(feats, self.fm, col if self.cross else 0)
        self.survey_bytes_per_impr = algorithmic_bytes_per_impression(feats, self.fm, 0)      # SURVEY 8d's own figure (C3: 2 600 B: "cross adds nothing if fused")
        self.granule_bytes_per_impr = granule_bytes_per_impression(feats, self.fm, col if self.cross else 0)
        self.bwd_bytes_per_impr = backward_bytes_per_impression(feats, self.fm)
        self.bytes_note = ("SURVEY 8d: per lookup 8 B id + 4D row read + 4D concat write"
                           + (", + 4 B FM logit per impression" if self.fm else "")
                           + (f" (= SURVEY 8d's figure for this workload: the gather alone, which `roofline.frac` uses); the fused launch also writes "
                              f"{4 * col} B per impression of cross output next to x: `roofline.frac_with_cross_write` counts them too" if self.cross else ""))
        # one bound call per id buffer: descriptors are built once, a step only enqueues the launch(es)
        # The output buffer is recycled every step, as torch's caching allocator does for any real
        # loop (the module path allocates `out` with torch.empty per call and gets the same block back).
        # Measured on MI355X for c2: recycled 62.4 us, 8 distinct output buffers 72.9 us.
        ld = 2 * col if self.cross else col
        self.ld = ld
        out = torch.empty((BATCH, ld), dtype=torch.float32, device=device)
        fmb = torch.empty((BATCH,), dtype=torch.float32, device=device) if self.fm else None
        # index checking stays ON in the timed path: the kernel records an out-of-range id in a device status word (no
        # cost when ids are in range); it is read once after the timed region (check_indices) -- the reference raises
        # IndexError per call, here the raise is deferred, not dropped
        self.calls = [ops.PreparedEmbed(self.plan, self.tables, ins, ws, out_ld=ld, out=out, fm=fmb, check_index=True)
                      for ins, ws in self.pool]
        self.fused = None
        if self.cross and ops.fused_cross_is_fast(self.plan) and os.environ.get("NRX_BENCH_FUSED_CROSS", "1") != "0":
            # one-launch gather -> cat[x, cross(x)] (ops.PreparedEmbedDcn, grouped kernel): 47.9 us vs 60.0 us for the two
            # launches below on MI355X; NRX_BENCH_FUSED_CROSS=0 measures the two-launch form
            self.fused = [ops.PreparedEmbedDcn(self.plan, self.tables, ins, self.cross_w, self.cross_b, out=out, check_index=True)
                          for ins, _ in self.pool]
            self.desc = self.desc.replace("(2 launches)", "(1 fused launch)")
        self.device = device
        from news_recsys_amd import _lib
        self.lib = _lib.load()
        # c5 is the WideDeep configuration: the headline launch is what WideDeep.get_inp_embedding runs -- the gather WITH the column routing
        # (src/model/sort/widedeep/model.py:53-69) -- and the plain concat of the same tables is the secondary leg (`plain_concat`);
        # NRX_BENCH_C5_PLAIN=1 swaps them back (rounds 1-3)
        self.plain_calls = None
        if wl == "c5" and os.environ.get("NRX_BENCH_C5_PLAIN") != "1":
            self.plain_calls = self.calls
            self.calls = self.wide_split_calls(check_index=True)
            self.desc += ("; headline = the gather with WideDeep.get_inp_embedding's column routing: column 0 of the 10 smallest tables -> wide "
                          "[B, 10], everything else -> deep [B, 1270] (row stride padded to 1280 floats = whole 128-byte lines, as the WideDeep model asks for)")

    def check_indices(self):
        for c in (self.fused if self.fused is not None else self.calls):
            c.check()

    def distinct_output_calls(self):
        """The same bound launches, each writing its own output buffer (8 x [B, ld]): what a pipeline that keeps every
        step's activations alive sees; the recycled single buffer (headline) stays in the 256 MiB Infinity Cache."""
        ops = self.ops
        outs = [torch.empty((BATCH, self.ld), dtype=torch.float32, device=self.device) for _ in self.pool]
        if self.fused is not None:
            return [ops.PreparedEmbedDcn(self.plan, self.tables, ins, self.cross_w, self.cross_b, out=o) for (ins, _), o in zip(self.pool, outs)]
        fmb = torch.empty((BATCH,), dtype=torch.float32, device=self.device) if self.fm else None
        return [ops.PreparedEmbed(self.plan, self.tables, ins, ws, out_ld=self.ld, out=o, fm=fmb) for (ins, ws), o in zip(self.pool, outs)]

    def wide_split_calls(self, check_index=False):
        """c5 only: the same gather with WideDeep.get_inp_embedding's column routing (src/model/sort/widedeep/model.py:53-69):
        column 0 of every wide feature goes to the wide tensor [B, n_wide], columns 1.. to the deep concat.  Wide features = the
        10 smallest tables (as in tests/test_full_size_baseline_shapes.py)."""
        ops = self.ops
        from news_recsys_amd._lib import NRX_SPARSE
        order = sorted(range(len(self.feats)), key=lambda i: self.feats[i]["rows"])
        wide_of = {i: k for k, i in enumerate(order[:10])}
        slots, col = [], 0
        for i, f in enumerate(self.feats):
            w = wide_of.get(i, -1)
            slots.append(ops.Slot(f["name"], NRX_SPARSE, self.plan.slots[i].table, f["dim"], 0, col, wide_col=w))
            col += f["dim"] - 1 if w >= 0 else f["dim"]
        plan = ops.EmbedPlan(slots, out_width=col, wide_width=len(wide_of))
        ld = (col + 31) // 32 * 32                  # 1270 -> 1280: rows start on a 128-byte line (WideDeep.get_inp_embedding pads the same way)
        out = torch.empty((BATCH, ld), dtype=torch.float32, device=self.device)
        return [ops.PreparedEmbed(plan, self.tables, ins, ws, out_ld=ld, out=out, check_index=check_index) for ins, ws in self.pool]

    def train_pass(self):
        """Forward (training form) + row-sparse backward of the gather path, bound once: (forward calls, backward calls).
        Upstream gradients are fixed random buffers (g_out [B, width], g_fm [B] for the FM workload)."""
        ops, dev = self.ops, self.device
        gen = torch.Generator(device=dev).manual_seed(7)
        g_out = torch.randn((BATCH, self.ld), device=dev, generator=gen)
        g_fm = torch.randn((BATCH,), device=dev, generator=gen) if self.fm else None
        out = torch.empty((BATCH, self.ld), dtype=torch.float32, device=dev)
        fmb = torch.empty((BATCH,), dtype=torch.float32, device=dev) if self.fm else None
        sums = torch.empty((BATCH, self.feats[0]["dim"]), dtype=torch.float32, device=dev) if self.fm else None
        fwd = [ops.PreparedEmbed(self.plan, self.tables, ins, ws, out_ld=self.ld, out=out, fm=fmb, fm_sums=sums)
               for ins, ws in self.pool[:2]]                    # two id batches alternate (the backward owns ~0.3 GB of buffers each)
        bwd = [ops.PreparedSparseBackward(f, g_out, g_fm) for f in fwd]
        return fwd, bwd

    @torch.no_grad()
    def step(self, i: int):
        if self.fused is not None:
            return self.fused[i % len(self.fused)].run()
        call = self.calls[i % len(self.calls)]
        res = call.run()
        if self.cross:       # (two-launch form) cat[x, cross(x)]: cross written next to x in the same [B, 2D] buffer
            buf, D = call.out, self.width
            rc = self.lib.nrx_dcn_v1_fwd(buf.data_ptr(), 2 * D, None, 0, BATCH, D, self.cross_w.shape[0], self.cross_w.data_ptr(),
                                         self.cross_b.data_ptr(), buf.data_ptr() + 4 * D, 2 * D,
                                         torch.cuda.current_stream(self.device).cuda_stream)
            assert rc == 0
        return res


# ------------------------------------------------------------------------------------ module path (the drop-in surface)
MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X dense fp32 matrix peak (guides/MI355X_MICROARCH.md); v_mfma_f32_32x32x2_f32


def dcn_v2_cross_leg(device, D: int):
    """The C3 shape's DCN-v2 cross layer (x_{l+1} = x0 * (W x_l + b) + x_l, the path's only dense contraction) on
    the matrix cores: one layer forward at B = 65 536 through the C-ABI (nrx_dcn_v2_layer_fwd), back-to-back launches after the clocks
    have settled, HIP events on the launch stream.  fp32 = the default, value-exact against the C oracle; bf16x3 = the opt-in split-bf16
    math (dcn_cfg.math), priced against the same fp32-equivalent flop count.  Algorithmic flops: (2 D^2 + 3 D) per row and layer."""
    from news_recsys_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device=device).manual_seed(20260116 + 3)
    x = torch.randn(BATCH, D, device=device, generator=gen)
    W = torch.randn(D, D, device=device, generator=gen) / D ** 0.5
    b = torch.randn(D, device=device, generator=gen) * 0.1
    out = torch.empty_like(x)
    lin = torch.empty_like(x)
    st = torch.cuda.current_stream(device).cuda_stream
    flops = (2.0 * D * D + 3.0 * D) * BATCH
    res = {"shape": f"one cross layer, B = {BATCH}, D = {D} (the C3 concat width), x0 == x_l, ReLU, inputs resident in HBM",
           "algorithmic_flops_per_launch": flops}
    for name, flags in (("fp32", 1), ("bf16x3", 3)):
        for train in (False, True):
            lp = lin.data_ptr() if train else None
            for _ in range(200):
                lib.nrx_dcn_v2_layer_fwd(x.data_ptr(), x.data_ptr(), D, BATCH, D, W.data_ptr(), b.data_ptr(), flags, out.data_ptr(), D, lp, st)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            n = 200
            for _ in range(n):
                lib.nrx_dcn_v2_layer_fwd(x.data_ptr(), x.data_ptr(), D, BATCH, D, W.data_ptr(), b.data_ptr(), flags, out.data_ptr(), D, lp, st)
            e.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(e) / n * 1e3
            tf = flops / (us * 1e-6) / 1e12
            if name == "fp32":
                roof = {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS}
            else:       # three bf16 MFMAs per fragment pair: 16 us of matrix time at D = 320 -- the launch is bound by memory, priced as such
                nbytes = (2 + (1 if train else 0)) * BATCH * D * 4 + D * D * 4 + D * 4      # x_l read, out (+ lin) written, W, b
                gbps = nbytes / (us * 1e-6) / 1e9
                roof = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                        "algorithmic_bytes_per_launch": nbytes, "fp32_equivalent_TFLOPs": tf}
            res[name + ("_training_form" if train else "")] = {"us": us, "roofline": roof}
    # forward + hand-written backward (prep + dgrad + wgrad) of the one layer through the autograd wrapper (ops.dcn_v2): three GEMMs of 2 D^2 B flops
    from news_recsys_amd import ops
    xg = x.clone().requires_grad_(True)
    Wg, bg = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    up = torch.randn(BATCH, D, device=device, generator=gen)
    for name in ("fp32", "bf16x3"):
        def step():
            torch.autograd.grad(ops.dcn_v2(xg, [Wg], [bg], math=name), [xg, Wg, bg], up)
        for _ in range(60):
            step()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 60
        for _ in range(n):
            step()
        e.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(e) / n * 1e3
        tf = 3 * 2.0 * D * D * BATCH / (us * 1e-6) / 1e12
        res[name + "_forward_backward"] = {"us": us, "fp32_equivalent_TFLOPs": tf, "frac_of_fp32_matrix_peak": tf / MFMA_F32_PEAK_TFLOPS,
                                           "flops": "3 x 2 D^2 B (forward, dgrad, wgrad)"}
    res["note"] = ("fp32: frac = algorithmic flops / launch time / the dense fp32 matrix peak.  bf16x3: the same contraction as three bf16 MFMAs per "
                   "fragment pair (fp32 accumulate) leaves ~16 us of matrix time at this shape, so the launch is priced against HBM by its algorithmic "
                   "bytes (x_l read once, out written once; profiles/r03_dcn_v2_bf16x3.txt says what it actually waits on); *_training_form also "
                   "writes x_l W^T + b for the backward; counters: profiles/r03_dcn_*_rocprof_summary.txt")
    return res


def module_path_leg(path: "SingleGpuPath", prepared_fb_ms, steps: int):
    """The same C2 work through the DROP-IN surface (what a user of the reference touches): the package's FM model class built from a
    train_cf_fm.yaml-shaped config with the bench's 26 x 1M x 16 tables -- FM.get_embeddings_from_batch / FM.forward(batch) on an input
    dict (src/model/BaseModel/base_model.py:284-308, src/model/sort/fm/model.py:48-59), eager, index check deferred, through
    torch.autograd with the fused row-sparse backward (embeddings.sparse_grad: fused).  Reported next to the bound (PreparedEmbed) path the
    headline times: GPU time per step from HIP events, host time per call from the wall clock of back-to-back un-synchronised calls."""
    import tempfile
    import yaml
    from news_recsys_amd import ops
    from news_recsys_amd.model.sort.fm.model import FM
    dev = path.device
    names = [f["name"] for f in path.feats]
    D, rows = path.feats[0]["dim"], path.feats[0]["rows"]
    cfg = {"name": "fm", "paths": {"out_basedir": tempfile.gettempdir(), "user_history_path": ""},
           "features": {"sparse_feature_names": names, "dense_feature_names": [], "array_feature_names": [], "item_feature_names": names[:13],
                        "user_feature_names": names[13:], "array_max_length": {}},
           "embeddings": {"embedding_size": {n: D for n in names}, "embedding_table_size": {n: rows for n in names}, "share_emb_table_features": {},
                          "sparse_grad": "fused"},
           "dataset": {"batch_size": BATCH, "num_workers": 0, "pin_memory": False},
           "train_hparams": {"val_freq": 1, "max_epoch": 1, "lr": 1e-3, "min_lr": 5e-6, "lr_milestones": [4, 20], "max_step": 30, "device": "gpu", "gpus": [0]}}
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(cfg, f)
    with torch.device(dev):
        model = FM(f.name)
    model = model.to(dev)
    for n, t in zip(names, path.tables):                     # the bench's own tables: no second 1.66 GB of parameters
        model.embedding_tables[n].weight.data = t
    os.unlink(f.name)
    fn = model.user_feature_names | model.item_feature_names
    out = {"model": "news_recsys_amd.model.sort.fm.model.FM from a train_cf_fm.yaml-shaped config (26 sparse x 1M x 16), input dict of int64 [B] ids, "
                    "eager, index check deferred, embeddings.sparse_grad: fused",
           "binding": "compiled (csrc/nrx_bind.cpp)" if ops._binding() is not None else "ctypes"}

    def timed(fn_, n, sync_each=False):
        for _ in range(5):
            fn_()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for _ in range(n):
            fn_()
        b.record()
        host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n, host * 1e3

    for B in (BATCH, 512):
        batches = [{nm: x[:B] if B == BATCH else x[:B].clone() for nm, x in zip(names, ins)} for ins, _ in path.pool[:2]]
        it = {"i": 0}

        def fwd_nograd():
            it["i"] += 1
            with torch.no_grad():
                model.get_embeddings_from_batch(batches[it["i"] & 1], fn)

        def fwd_bwd():
            it["i"] += 1
            p = model(batches[it["i"] & 1])                   # sigmoid(bias + FM logit): gather + FM epilogue, autograd node
            p.sum().backward()                                # -> FM gradient folded into the row-sparse backward, results in the sink
            if model._sparse_sink is not None:
                model._sparse_sink.clear()                    # (an optimizer would consume them here)

        def fwd_bwd_hook():                                   # the step as a Lightning trainer drives it: LightningModule.backward(loss)
            it["i"] += 1
            p = model(batches[it["i"] & 1])
            model.backward(p.sum())
            if model._sparse_sink is not None:
                model._sparse_sink.clear()

        g_ms, h_ms = timed(fwd_nograd, min(steps, 200))
        fb_ms, fbh_ms = timed(fwd_bwd, min(steps, 100))
        fbk_ms, fbkh_ms = timed(fwd_bwd_hook, min(steps, 100))
        with torch.autograd.set_multithreading_enabled(False):      # backward nodes run on the calling thread: no hand-off to the engine's device thread
            fb1_ms, fb1h_ms = timed(fwd_bwd, min(steps, 100))
        # the same step replayed from a HIP graph (news_recsys_amd.graph.GraphedStep: index check still ON -- deferred status word inside
        # the graph, IndexError at the next call --, plan made inline in the captured backward, count read on the device)
        graphed = None
        try:
            from news_recsys_amd.graph import GraphedStep

            def gstep(bt):
                p = model(bt)
                p.sum().backward()
                if model._sparse_sink is not None:
                    model._sparse_sink.clear()
                return p

            gs = GraphedStep(gstep, batches[0], warmup=3)

            def graphed_call():
                it["i"] += 1
                gs(batches[it["i"] & 1])

            gr_ms, grh_ms = timed(graphed_call, min(steps, 100))
            rp_ms, rph_ms = timed(gs.graph.replay, min(steps, 100))
            gs.check()
            graphed = {"forward_backward_graphed_us": gr_ms * 1e3, "forward_backward_graphed_host_us_per_step": grh_ms * 1e3,
                       "replay_only_us": rp_ms * 1e3, "replay_only_host_us_per_step": rph_ms * 1e3}
            del gs
        except Exception as e:       # noqa: BLE001 -- a secondary figure must not cost the line
            graphed = {"error": f"{type(e).__name__}: {e}"[:300]}
        train = None
        if B == 512:
            # a whole training step of the embedding layer at a batch size the reference's YAMLs use: forward + backward (LightningModule.backward hook) +
            # optimizer -- `fused`: FusedSparseAdam on the looked-up rows (sparse_grad: fused; the one-launch row-sparse backward feeds its sink),
            # `exact`: the reference's dense AdamW over every row of the 26 x 1M-row tables, streamed once from the same sink (sparse_grad: exact)
            try:
                from news_recsys_amd.model.model_utils.optim import ExactDenseAdamW, FusedSparseAdam
                tabs = [model.embedding_tables[n].weight for n in names]
                train = {}
                for mode in ("fused", "exact"):
                    opt = (FusedSparseAdam(model._sparse_sink, lr=1e-6, params=tabs) if mode == "fused" else
                           ExactDenseAdamW(model._sparse_sink, tabs, lr=1e-6, weight_decay=0.0))     # (tiny lr, no decay: the bench's tables stay what they are)

                    def train_step():
                        it["i"] += 1
                        p = model(batches[it["i"] & 1])
                        model.backward(p.sum())
                        opt.step()

                    t_ms, th_ms = timed(train_step, 100 if mode == "fused" else 20)
                    train[mode] = {"step_us": t_ms * 1e3, "host_us_per_step": th_ms * 1e3}
                    del opt
                    torch.cuda.empty_cache()
            except Exception as e:       # noqa: BLE001 -- a secondary figure must not cost the line
                train = {"error": f"{type(e).__name__}: {e}"[:300]}
        out[f"B{B}"] = {"get_embeddings_from_batch_us": g_ms * 1e3, "get_embeddings_from_batch_host_us_per_call": h_ms * 1e3,
                        "graphed": graphed, **({"train_step": train} if train is not None else {}),
                        "forward_backward_autograd_us": fb_ms * 1e3, "forward_backward_host_us_per_step": fbh_ms * 1e3,
                        "forward_backward_lightning_hook_us": fbk_ms * 1e3, "forward_backward_lightning_hook_host_us_per_step": fbkh_ms * 1e3,
                        "forward_backward_autograd_us_engine_thread_off": fb1_ms * 1e3,
                        "forward_backward_host_us_per_step_engine_thread_off": fb1h_ms * 1e3}
    if prepared_fb_ms:
        e = out["B%d" % BATCH]
        e["prepared_forward_backward_us"] = prepared_fb_ms * 1e3
        e["autograd_over_prepared"] = e["forward_backward_autograd_us"] / (prepared_fb_ms * 1e3)
        e["autograd_over_prepared_engine_thread_off"] = e["forward_backward_autograd_us_engine_thread_off"] / (prepared_fb_ms * 1e3)
        e["lightning_hook_over_prepared"] = e["forward_backward_lightning_hook_us"] / (prepared_fb_ms * 1e3)
    out["note"] = ("GPU time per step from HIP events around back-to-back calls (host-bound when it equals the host time); the autograd step = "
                   "FM.forward(batch) + p.sum().backward() with the row-sparse gradients left in the model's SparseGradSink (planning on the side "
                   "stream at forward time); `prepared` = the bound launches of the fwd_bwd leg; *_lightning_hook = the step with the backward entered "
                   "through the model's LightningModule.backward hook (what a Lightning trainer calls; it runs the nodes on the calling thread); "
                   "*_engine_thread_off = the plain step under "
                   "torch.autograd.set_multithreading_enabled(False): PyTorch's backward otherwise hands every step to its device thread, ~140 us "
                   "of wake-up and GIL hand-over per step on this host (tools/host_profile_module_step.py); graphed = the same forward + backward "
                   "captured once in a HIP graph (GraphedStep) and replayed: forward_backward_graphed_us includes the copy of the batch's 26 id tensors into "
                   "the graph's static inputs, replay_only_us is the graph alone; the out-of-range-id check stays on inside the graph (deferred); "
                   "B512.train_step = forward + backward (hook) + optimizer step, eager: `fused` = FusedSparseAdam on the looked-up rows, `exact` = "
                   "the reference's dense AdamW over every row of the 26 x 1M-row tables (ExactDenseAdamW: 10 GB of table traffic per step by definition)")
    ops.flush_index_checks()
    del model
    return out


# ------------------------------------------------------------------------------------ input pipeline (host -> device)
def input_pipeline_leg(path: "SingleGpuPath", n_batches: int = 24):
    """SURVEY 8f row 1 at the headline shape: the columnar loader (news_recsys_amd.dataset.DataReader.columnar.ColumnarLoader, what replaces the
    reference's text DataReader + collate, src/dataset/DataReader/data_reader.py:54-115) streaming C2 batches -- 26 int64 id columns, B = 65 536 --
    from PAGE-LOCKED host columns into device batches, the next batch's host -> device copies on a side stream under the current batch's forward
    launch.  Reported next to `value`, never as it: samples/s, the bytes that crossed the host link per second, and that rate as a fraction of what
    one large pinned copy reaches on this box in this run."""
    import numpy as np
    from news_recsys_amd import ops
    from news_recsys_amd.dataset.DataReader.columnar import ColumnarDataset, ColumnarLoader
    dev = path.device
    names = [f["name"] for f in path.feats]
    rows = [f["rows"] for f in path.feats]
    n = n_batches * BATCH
    rng = np.random.default_rng(7)
    cols = {nm: rng.integers(1, r, n, dtype=np.int64) for nm, r in zip(names, rows)}
    ds = ColumnarDataset.from_arrays(cols, np.zeros((n, 1), np.float32))
    loader = ColumnarLoader(ds, BATCH, dev, pinned=True)
    # the link: one 256 MB page-locked buffer, copied asynchronously, a few times
    big = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty_like(big, device=dev)
    for _ in range(2):
        dst.copy_(big, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        dst.copy_(big, non_blocking=True)
    torch.cuda.synchronize()
    link = 4 * big.numel() / (time.perf_counter() - t0) / 1e9
    del big, dst

    def epoch(consume):
        for b in loader:
            if consume:
                with torch.no_grad():
                    ops.embed_apply(path.plan, path.tables, [b[nm] for nm in names], [None] * len(names), index_check="off")
    epoch(True)                                   # pins the columns, warms the allocator
    torch.cuda.synchronize()
    res = {}
    for consume in (False, True):
        t0 = time.perf_counter()
        epoch(consume)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nbytes = n * (len(names) * 8 + 4)
        res["with_forward" if consume else "copies_only"] = {"samples_per_s": n / dt, "pcie_GBps": nbytes / dt / 1e9, "frac_of_link": nbytes / dt / 1e9 / link,
                                                             "ms_per_batch": dt / n_batches * 1e3}
    return {**res["with_forward"], "copies_only": res["copies_only"], "link_GBps_one_large_pinned_copy": link, "batches": n_batches,
            "bytes_per_sample": len(names) * 8 + 4,
            "note": "ColumnarLoader(pinned=True): the dataset's columns page-locked once, batch-blocked ([batch][column][B]): every batch = ONE 13.6 MB asynchronous "
                    "copy (+ the labels) on a side stream, double-buffered -- 27 copies of 512 KB per batch ran at 36 % of the link; with_forward = each batch also runs the C2 gather + FM launch (ops.embed_apply, no grad) on the main stream.  "
                    "The mmap form of the loader (two host copies per batch) reached 10 M samples/s (profiles/r01_loader_throughput.txt)"}


# ------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(path: SingleGpuPath, budget_s: float = 12.0):
    """Times the CPU oracle's C/OpenMP restatement of the reference path (oracle/nrx_oracle.c, checked
    against the same goldens as oracle/ref_np.py; 'port') on ALL host cores, on a bounded sample of the
    same workload: same tables (copied to host), same id batches, gather(+pool)->concat (+FM / cross)."""
    from oracle import ref_c
    tnames = []
    for f in path.feats:
        t = f.get("share", f["name"])
        if t not in tnames:
            tnames.append(t)
    tables = {n: t.cpu().numpy() for n, t in zip(tnames, path.tables)}
    ins, ws = path.pool[0]
    Bs = BATCH
    feats = []
    for f, x, w in sorted(zip(path.feats, ins, ws), key=lambda t: t[0]["name"]):
        kind = ref_c.BAG_MASKED_MEAN if f["bag"] else ref_c.SPARSE
        feats.append(dict(kind=kind, table=tables[f.get("share", f["name"])], index=x[:Bs].cpu().numpy(),
                          weight=None if w is None else w[:Bs].cpu().numpy()))
    call = ref_c.EmbedCall(feats, Bs)
    cores = ref_c.threads()
    cw = path.cross_w.cpu().numpy() if path.cross else None
    cb = path.cross_b.cpu().numpy() if path.cross else None

    def one():
        out = call.run()
        if path.fm:
            ref_c.fm_logit(out, len(call.dims), call.dims[0])
        if path.cross:
            ref_c.dcn_v1(out, cw, cb)

    one()
    t0 = time.perf_counter()
    reps = 0
    while True:
        one()
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 2000:
            break
    dt = (time.perf_counter() - t0) / reps
    return {"value": Bs / dt, "unit": "impressions/s", "cores": cores, "kind": "port",
            "sample": f"{reps} passes of {Bs} impressions of the same workload (oracle/nrx_oracle.c, OpenMP, "
                      f"{cores} threads); {dt * 1e3:.2f} ms/pass; host has {os.cpu_count()} logical cores"}


def cpu_baseline_torch(path: SingleGpuPath, budget_s: float = 10.0):
    """The reference's own module code restated in stock PyTorch on the host cores: one
    F.embedding per feature (get_feature_embedding, base_model.py:262-271), masked-mean pooling (:273-282), torch.cat
    (:308), then the model's interaction -- FM from sums (fm/model.py:18-26) or the DCN cross.  For DCN the reference
    materialises a [B, D, D] outer product per layer (dcn_arch.py:25): timed at B = 8192 and labelled (26.8 GB at
    B = 65536); the rest runs at the full batch.  torch.set_num_threads(all cores)."""
    import torch.nn.functional as F
    cores = os.cpu_count() or 1
    B = 8192 if path.cross else BATCH
    tnames = []
    for f in path.feats:
        t = f.get("share", f["name"])
        if t not in tnames:
            tnames.append(t)
    tables = {n: t.cpu() for n, t in zip(tnames, path.tables)}
    ins, ws = path.pool[0]
    ins = [x[:B].cpu() for x in ins]
    ws = [None if w is None else w[:B].cpu() for w in ws]
    cw = path.cross_w.cpu() if path.cross else None
    cb = path.cross_b.cpu() if path.cross else None
    D0 = path.feats[0]["dim"]

    @torch.no_grad()
    def one():
        cols = []
        for f, x, w in zip(path.feats, ins, ws):
            e = F.embedding(x.long(), tables[f.get("share", f["name"])], padding_idx=0)
            if f["bag"]:
                e = (e * w.unsqueeze(-1)).sum(dim=1) / (w.sum(dim=1, keepdim=True) + 1e-8)
            cols.append(e)
        x = torch.cat(cols, dim=1)
        if path.fm:
            fv = x.view(B, len(cols), D0)
            v = fv[:, :, 1:]
            return fv[:, :, 0].sum(1) + 0.5 * (v.sum(1).pow(2) - v.pow(2).sum(1)).sum(1)
        if path.cross:
            x0, xl = x.unsqueeze(-1), x.unsqueeze(-1)
            for l in range(cw.shape[0]):
                xl = torch.matmul(torch.matmul(x0, xl.transpose(1, 2)), cw[l].unsqueeze(-1)) + cb[l].unsqueeze(-1) + xl
            return torch.cat([x, xl.squeeze(-1)], dim=1)
        return x

    # torch's intra-op pool does not scale to hundreds of threads on these small ops (256 threads measured 100x slower
    # than 32 on the MI355X host): time a few pool sizes within the budget and report the best one, with its size
    best, best_threads, reps = 1e30, 1, 0
    t0 = time.perf_counter()
    for nt in sorted({min(cores, n) for n in (32, 64, 16, cores)}, key=lambda n: (n == cores, n)):
        if time.perf_counter() - t0 > budget_s:
            break
        torch.set_num_threads(nt)
        one()
        t_nt = time.perf_counter()
        while True:
            t1 = time.perf_counter()
            one()
            dt1 = time.perf_counter() - t1
            reps += 1
            if dt1 < best:
                best, best_threads = dt1, nt
            if time.perf_counter() - t_nt > budget_s / 4 or reps >= 400:
                break
    return {"value": B / best, "unit": "impressions/s", "cores": best_threads, "kind": "torch-restatement",
            "sample": f"best of {reps} passes of {B} impressions, stock PyTorch CPU eager restatement of the reference module code "
                      f"(F.embedding per feature + cat + interaction), best intra-op pool = {best_threads} threads of {cores} "
                      f"logical cores; {best * 1e3:.1f} ms/pass"
                      + ("; DCN in the reference's [B,D,D] outer-product form, hence B = 8192" if path.cross else "")}


# ------------------------------------------------------------------------------------ main
def _direct_1gpu_reference(workload: str):
    """The direct single-GPU path's headline value of `workload` from the newest committed bench lines (profiles/rNN_bench_lines_c2_c3_c4_c5.jsonl)."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_lines_c2_c3_c4_c5.jsonl")), reverse=True):
        try:
            for ln in open(fn):
                d = json.loads(ln)
                if d.get("n_gpus") == 1 and str(d.get("config", {}).get("workload", "")).startswith(workload + ":"):
                    return {"value": d["value"], "source": os.path.relpath(fn, ROOT)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def main():
    args = parse_args()
    if args.shard_mode == "default":
        # the N > 1 headline is the layout north_star scales: EVERY table row-sharded, ids and rows crossing the fabric (round 5 took the planner
        # layout for c2 / c3, whose 64 MB tables it replicates -- zero all-to-all: a data-parallel copy would have "scaled" 8x)
        args.shard_mode = "row"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # (a plain `python bench.py --gpus N` never gets here: self_launch() above started the ranks)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`")
    # NRX_BENCH_HOST_STAGED=1 (testing): all ranks share GPU 0 and exchange over gloo through host-staged buffers, so the
    # N > 1 control flow and the sharded data path can be run on a one-GPU box; the numbers it prints are not a measurement
    staged = os.environ.get("NRX_BENCH_HOST_STAGED") == "1"
    if staged:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from news_recsys_amd import ops
    ops.set_index_check("off")            # ids are generated in range; the flag read-back would sync every step

    dist = None
    if world > 1 or os.environ.get("NRX_BENCH_FORCE_DIST") == "1":       # env: exercise the RCCL plumbing on one GPU (testing)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if staged:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    seed = 20260116 + {"c2": 2, "c3": 3, "c4": 4, "c5": 5}[args.workload] + rank
    if world == 1 and not args.force_sharded:
        path = SingleGpuPath(args.workload, device, seed, id_dist=args.ids)
        if args.ids != "uniform":
            path.desc = path.desc.replace("uniform ids", "ids") + f" -- ids ~ {args.ids}(1.05)"
        step = path.step
        bytes_per_impr = path.bytes_per_impr
        desc = path.desc
        parallelism = "single-gpu"
    else:
        from news_recsys_amd.sharding import ShardedBenchPath
        path = ShardedBenchPath(args.workload, device, seed, rank, world, BATCH, args.shard_mode, host_staged=staged)
        step = path.step
        bytes_per_impr = path.bytes_per_impr
        desc = path.desc
        parallelism = (f"batch-parallel {BATCH}/GPU over {world} GPUs; tables: {path.n_sharded} row-sharded (RCCL all-to-all "
                       f"id routing + row return), {path.n_replicated} replicated (planner mode={args.shard_mode}: tables "
                       f"<= 256 MiB are replicated)")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn):
        for i in range(args.warmup):
            step_fn(i)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()
        for i in range(args.steps):
            step_fn(args.warmup + i)
        e1.record()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, e0.elapsed_time(e1) / args.steps

    for i in range(args.warmup):
        step(i)
    barrier()
    # HIP events on the launch stream (torch's current stream) bracket the timed region: mean launch
    # duration of the step's kernel(s) = elapsed / steps (back-to-back launches, gaps included)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    e0.record()
    for i in range(args.steps):
        step(args.warmup + i)
    e1.record()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    kern_ms = e0.elapsed_time(e1) / args.steps

    # per-step spread (outside the timed region, rank 0 / N=1 only): each step bracketed by its own pair of events
    spread = None
    if world == 1:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        for i, (a, b) in enumerate(evs):
            a.record()
            step(args.warmup + args.steps + i)
            b.record()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        spread = {"p10": us[6], "p50": us[32], "p90": us[57], "n": 64,
                  "note": "single steps between their own HIP events (includes one launch gap each); not part of the timed region"}

    if hasattr(path, "check_indices"):
        path.check_indices()              # deferred IndexError of the timed launches (the reference raises per call)

    def _traffic(key):
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                return json.load(f).get(key, {}).get("hbm_bytes_per_launch")
        except Exception:       # noqa: BLE001
            return None

    # what a plain copy sustains on THIS box, same run: 1 GiB read + 1 GiB written per launch, 16 bytes per lane
    stream_copy = None
    if world == 1 and not args.force_sharded and hasattr(path, "lib"):
        try:
            nbytes = 1 << 30
            src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
            dst = torch.empty_like(src)
            st = torch.cuda.current_stream(device).cuda_stream
            for _ in range(3):
                path.lib.nrx_stream_copy(dst.data_ptr(), src.data_ptr(), nbytes, st)
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(10):
                path.lib.nrx_stream_copy(dst.data_ptr(), src.data_ptr(), nbytes, st)
            b_.record()
            torch.cuda.synchronize()
            stream_copy = 2 * nbytes * 10 / (a_.elapsed_time(b_) * 1e-3) / 1e9
            del src, dst
            torch.cuda.empty_cache()
        except Exception:       # noqa: BLE001
            stream_copy = None

    # secondary legs at N = 1, outside the headline timed region
    distinct = fwd_bwd = wide_split = module_path = dcn_v2_cross = None
    def sharded_train_leg(path):
        # the sharded engine's TRAINING step (every world size, 1 included): forward exchange + gradient all-to-all back to the owners + owner-side
        # scatter into the shards' dense gradients, through autograd -- eager, un-bound (no PreparedShardedForward counterpart for the backward)
        fwd_bwd = None
        try:
            if path.train_setup():
                nst = max(5, min(args.steps, 30))
                for i in range(3):
                    path.train_step(i)
                barrier()
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                ea.record()
                for i in range(nst):
                    path.train_step(3 + i)
                eb.record()
                barrier()
                dts = time.perf_counter() - t0
                if dist is not None:
                    tt = torch.tensor([dts], dtype=torch.float64, device=device)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dts = tt.item()
                gpu_ms = ea.elapsed_time(eb) / nst
                fb_bytes = (path.bytes_per_impr + (4 * path.feats[0].dim if path.fm else 0) + path.bwd_bytes_per_impr) * BATCH      # per rank
                fb_ach = fb_bytes / (dts / nst) / 1e9
                bound = getattr(path, "engine", "legacy") == "feat"
                fwd_bwd = {"ms_per_step": dts * 1e3 / nst, "gpu_ms_per_step_rank0": gpu_ms, "value": BATCH * world * nst / dts,
                           "unit": "impressions/s", "steps": nst, "engine": getattr(path, "engine", "legacy"),
                           "host_bound": bool(dts * 1e3 / nst > 1.15 * gpu_ms),
                           "roofline": {"bound": "hbm", "achieved": fb_ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fb_ach / HBM_PEAK_GBPS,
                                        "algorithmic_bytes_per_step_per_rank": fb_bytes,
                                        "note": "the DIRECT path's algorithmic bytes of one rank's batch (forward in training form + row-sparse backward, "
                                                "bench.backward_bytes_per_impression) / wall time per step, max over ranks: the exchange's own traffic (ids, "
                                                "rows and gradient rows through the send / receive buffers) is overhead of the method, not counted"},
                           "mode": ("shard_step.PreparedShardedStep, bound: nrx_route_feat (one launch) + all-to-all of owner ids (+ positions) + the owner's "
                                    "placing gather straight into the requester's concat (one-sided; FM: one pass over the finished concat) -- bag features: "
                                    "nrx_route_bags_runs (routing + weight normalisation + run bounds in one launch) + owner-side pooling; backward: the owner "
                                    "plans first (the single-GPU planners), its plan's dest[] travels back, nrx_embed_bwd_scatter(_multi) writes every upstream "
                                    "row (FM term folded in) to its place in the owner's gradient arena, the owner's sorted walk reduces the listed rows into "
                                    "ROW-SPARSE (keys, values): deterministic, no dense shard gradient, no float atomic, nothing read back; independent "
                                    "exchange groups run side by side on two streams; at world 1 the all-to-alls vanish (receive buffers = send buffers)") if bound else
                                   ("sharded engine, autograd form (RowShardedEmbedding.forward + backward): id routing + all-to-alls + owner gather / "
                                    "owner-side pooling + fused final launch; backward: slot scatter of the upstream rows, gradient all-to-all to the owners, "
                                    "owner-side scatter-add into the shards' DENSE gradients (zero-filled every step); eager launches from Python "
                                    "(host time included: wall clock, max over ranks)")}
                if bound and world == 1:
                    # the same step captured in a HIP graph (it allocates nothing, reads nothing back and forks / joins its groups with wait_stream):
                    # GPU time without the host's launches; one id set (the eager loop above alternates two)
                    try:
                        c = path.calls[0]
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph):
                            c.run()
                            c.backward()
                        for _ in range(3):
                            graph.replay()
                        ga, gb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        ga.record()
                        for _ in range(nst):
                            graph.replay()
                        gb.record()
                        torch.cuda.synchronize()
                        fwd_bwd["graphed"] = {"replay_ms_per_step": ga.elapsed_time(gb) / nst,
                                              "note": "forward + backward of the bound step captured in one HIP graph, replayed: same launches, no host time -- and NOT faster "
                                                      "than the eager loop while the host keeps ahead of the GPU (a graph's nodes are dispatched one by one with a "
                                                      "dependency barrier between them; eager launches queue back to back): what the capture buys is a step whose "
                                                      "time does not depend on the host"}
                        del graph
                    except Exception as e:          # noqa: BLE001
                        fwd_bwd["graphed"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            else:
                fwd_bwd = {"skipped": "the shards' dense gradients (one zero-filled [local rows, dim] tensor per table and step) do not fit next to the tables"}
        except Exception as e:          # noqa: BLE001 -- a secondary leg must not cost the headline
            fwd_bwd = {"error": f"{type(e).__name__}: {e}"[:300]}
        return fwd_bwd

    input_pipeline = None
    if world == 1 and not args.force_sharded and not args.headline_only and args.workload == "c2" and hasattr(path, "plan"):
        try:
            input_pipeline = input_pipeline_leg(path)
        except Exception as e:          # noqa: BLE001 -- a secondary leg must not cost the line
            input_pipeline = {"error": f"{type(e).__name__}: {e}"[:300]}
    if world == 1 and hasattr(path, "train_setup") and not args.headline_only:
        fwd_bwd = sharded_train_leg(path)        # (N > 1: behind the headline line and the watchdog, with the other secondary legs)
    if world == 1 and not args.force_sharded and not args.headline_only:
        def time_calls(fn, n):
            # warm-up by TIME (>= 50 ms of the same launches): the legs that contain latency-bound launches (the backward's
            # planning) depend on the core clock, which takes tens of milliseconds to come up after an idle gap
            # (profiles/r02_dcn_v2_phase_probe.txt section 0); the HBM-bound headline reads the same either way
            t_w = time.perf_counter()
            i = 0
            while i < min(10, args.warmup) or (time.perf_counter() - t_w < 0.05 and args.warmup > 0):
                fn(i)
                i += 1
                if i % 8 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(n):
                fn(i)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n

        steps2 = min(args.steps, 100)
        if path.tables and sum(t.numel() for t in path.tables) * 4 < (200 << 30):     # room for 8 more output buffers
            dcalls = path.distinct_output_calls()
            ms = time_calls(lambda i: dcalls[i % len(dcalls)].run(), steps2)
            ach = bytes_per_impr * BATCH / (ms * 1e-3) / 1e9
            distinct = {"kernel_ms_mean": ms, "achieved": ach, "frac": ach / HBM_PEAK_GBPS, "unit": "GB/s",
                        "note": "same launches, 8 distinct output buffers instead of one recycled buffer (the recycled buffer "
                                "stays in the 256 MiB Infinity Cache)"}
            del dcalls
        if args.workload == "c5":
            if getattr(path, "plain_calls", None) is not None:     # headline = the split; this leg = the plain concat of the same tables
                wcalls, key_note = path.plain_calls, ("plain_concat", "the same tables and ids WITHOUT the Wide&Deep column routing: one [B, 1280] concat "
                                                      "(the headline of rounds 1-3)")
            else:
                wcalls, key_note = path.wide_split_calls(), ("wide_split", "the same tables and ids through WideDeep.get_inp_embedding's column routing "
                                                             "(widedeep/model.py:53-69): column 0 of the 10 smallest tables -> wide tensor [B, 10], the rest -> "
                                                             "deep concat [B, 1270], row stride 1280")
            ms = time_calls(lambda i: wcalls[i % len(wcalls)].run(), steps2)
            ach = bytes_per_impr * BATCH / (ms * 1e-3) / 1e9
            wide_split = {"leg": key_note[0], "kernel_ms_mean": ms, "achieved": ach, "frac": ach / HBM_PEAK_GBPS, "unit": "GB/s", "note": key_note[1]}
            del wcalls
        if not path.cross:
            fwd, bwd = path.train_pass()
            f_ms = time_calls(lambda i: fwd[i % len(fwd)].run(), steps2)
            def both(i):
                # planning inline, after the forward: with nothing between forward and backward to hide it behind, running the
                # planner on a side stream NEXT to the forward (PreparedSparseBackward.plan_ahead, what a training step with a
                # dense model in between does) only makes the two contend: C2 283 vs 267 us, C4 384 vs 364 (tools/profile_fwd_bwd.py)
                fwd[i % len(fwd)].run()
                bwd[i % len(bwd)].run()
            fb_ms = time_calls(both, steps2)
            # the leg is not only timed: every group's plan must list each (table, row) once, ascending, with segments that tile
            # the lookups exactly (a mis-sorted plan would still run at full speed)
            planners = []
            fwd[0].run()          # (the concat and the FM field sums the backward folds in must be THIS batch's: the output buffer is recycled over the id pool)
            for g in bwd[0].run():
                nu = int(g["counts"][0].item())
                uq, sg = g["uniq"][:nu], g["seg"][:nu + 1]
                planners.append("one-kernel (LDS bitmaps)" if g["pairs"] else "sorted")
                if g["pairs"]:
                    # the one-kernel planner defines segments for the walk rows only: its result is checked against the SORTED planner's on the
                    # same ids instead -- same unique rows, same row gradients bit for bit
                    keys, vals = uq.clone(), g["values"][:nu].clone()
                    pol, g["policy"] = g["policy"], None
                    bwd[0].run()
                    g["policy"] = pol
                    nu2 = int(g["counts"][0].item())
                    sg = g["seg"][:nu2 + 1]
                    if not (nu2 == nu and torch.equal(keys, g["uniq"][:nu]) and torch.equal(vals.view(torch.int32), g["values"][:nu].view(torch.int32))):
                        raise SystemExit("fwd_bwd leg: the one-kernel planner's gradients differ from the sorted planner's")
                if not (nu > 0 and bool(torch.all(uq[1:] > uq[:-1])) and int(sg[0].item()) == 0 and int(sg[-1].item()) == g["total"]
                        and bool(torch.all(sg[1:] > sg[:-1])) and bool(torch.isfinite(g["values"][:nu]).all())):
                    raise SystemExit("fwd_bwd leg: the row-sparse backward's plan is inconsistent (unique rows not ascending / segments "
                                     "do not tile the lookups)")
            fb_bytes = (bytes_per_impr + (4 * path.feats[0]["dim"] if path.fm else 0) + path.bwd_bytes_per_impr) * BATCH
            fb_ach = fb_bytes / (fb_ms * 1e-3) / 1e9
            fwd_bwd = {"ms_per_step": fb_ms, "value": BATCH / (fb_ms * 1e-3), "unit": "impressions/s",
                       "forward_ms": f_ms, "backward_ms": fb_ms - f_ms, "planner": planners,
                       "roofline": {"bound": "hbm", "achieved": fb_ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fb_ach / HBM_PEAK_GBPS,
                                    "algorithmic_bytes_per_step": fb_bytes,
                                    "algorithmic_bytes_definition": "forward (training form: + 4D field sums written per sample for FM) + backward: per lookup 8 B id (planner) "
                                                                    "+ 4D upstream row read (+ 4 B weight for a bag lookup) (+ 4D forward value read for FM fields) + 4D gradient "
                                                                    "row written + 8 B key written; per sample 4D field sums + 4 B g_fm read (FM).  The sort's own traffic is not counted",
                                    "traffic": _traffic(args.workload + "_fwd_bwd") if args.ids == "uniform" else None, "traffic_unit": "bytes/step",
                                    "note": "achieved = algorithmic bytes per step / mean step time (HIP events over the timed steps: forward + planning + "
                                            "reduction, ~10 launches); traffic = fabric bytes per step summed over those launches from the committed PMC passes"},
                       "mode": "forward (training form: + FM field sums) + deterministic row-sparse backward: the planner named in `planner` ("
                               + ("nrx_sparse_plan_lds: one kernel, row bitmaps in LDS, pair records for the rows looked up twice" if "one-kernel (LDS bitmaps)" in planners
                                  else "nrx_sparse_plan_place / _ex: table-segmented stable radix sort of the row bits, unique rows, segments, placement")
                               + "; inline here, chosen per batch from the previous batch's duplicate statistics) + the placement pass + the sorted walk with the FM "
                               "gradient folded in; upstream gradients g_out [B, width]" + (" and g_fm [B]" if path.fm else "") + " given; result = unique "
                               "(table,row) keys + summed row gradients on the device (what optim.FusedSparseAdam consumes)"}
            del fwd, bwd
        if args.workload == "c3":
            try:
                dcn_v2_cross = dcn_v2_cross_leg(device, sum(f["dim"] for f in path.feats))
            except Exception as e:          # noqa: BLE001 -- a secondary leg must not cost the headline
                dcn_v2_cross = {"error": f"{type(e).__name__}: {e}"}
        if args.workload == "c2" and args.ids == "uniform" and os.environ.get("NRX_BENCH_MODULE_PATH", "1") != "0":
            try:
                module_path = module_path_leg(path, fwd_bwd["ms_per_step"] if fwd_bwd else None, steps2)
            except Exception as e:          # noqa: BLE001 -- a secondary leg must not cost the headline
                module_path = {"error": f"{type(e).__name__}: {e}"}

    if hasattr(path, "overflowed"):
        # every rank must take the same branch: agree on the flag first (a lone SystemExit would strand the peers in the
        # next collective until the RCCL timeout)
        over = torch.tensor([1 if path.overflowed() else 0], dtype=torch.int32, device=device)
        if dist is not None:
            if staged:
                oc = over.cpu()
                dist.all_reduce(oc, op=dist.ReduceOp.MAX)
                over = oc
            else:
                dist.all_reduce(over, op=dist.ReduceOp.MAX)
        if int(over.item()):
            if dist is not None:
                dist.barrier()
                dist.destroy_process_group()
            raise SystemExit("fixed-capacity exchange overflowed on some rank: rerun with a larger slack (ids too skewed)")
    planner = strong = a2a = sharded_fb = None
    secondary_note = None
    emitted = {"done": False}
    def _build_line():
        total_impr = BATCH * world * args.steps
        ms_per_step = dt * 1e3 / args.steps
        # a workload with the fused cross (c3): SURVEY 8d prices the launch by the gather alone (2 600 B per impression, "cross adds nothing
        # if fused") -- that is `frac`; the launch also WRITES the cross half of cat[x, cross] (4 x width B per impression), which
        # `frac_with_cross_write` counts as well (the definition this line used up to round 3)
        survey_bytes = getattr(path, "survey_bytes_per_impr", bytes_per_impr)
        achieved = survey_bytes * BATCH / (kern_ms * 1e-3) / 1e9
        achieved_cw = bytes_per_impr * BATCH / (kern_ms * 1e-3) / 1e9
        traffic = _traffic(args.workload) if (world == 1 and args.ids == "uniform") else None
        granule = getattr(path, "granule_bytes_per_impr", None)
        out = {
            "metric": "impressions/sec at batch 65536 (embedding hot path forward)",
            "value": total_impr / dt,
            "unit": "impressions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "headline_definition_version": 2,     # 2 (round 4 on): c5 `value` = the Wide&Deep split launch at row stride 1280 (rounds 1-3: the plain concat, now
                                                  # the `plain_concat` leg); c3 roofline.frac prices the gather alone (2 600 B), frac_with_cross_write the old bytes
            "config": {"workload": desc, "batch_per_gpu": BATCH, "parallelism": parallelism,
                       "algorithmic_bytes_per_impression": survey_bytes,
                       **({"algorithmic_bytes_per_impression_with_cross_write": bytes_per_impr} if survey_bytes != bytes_per_impr else {}),
                       "algorithmic_bytes_definition": getattr(path, "bytes_note", "SURVEY 8d"), "id_pool": 8,
                       "output_buffer": "recycled each step (as the caching allocator does); `distinct_output_buffers` has the other mode",
                       "index_check": "on (device status word, read once after the timed region)"},
            "step_us": spread,
            # (algorithmic bytes that exceed what the HBM peak could deliver -- skewed ids over a table that fits the 256 MB Infinity Cache / the L2s:
            # most row reads never reach DRAM -- are not an HBM-roofline figure: the line says so instead of printing a fraction above 1 as "hbm")
            "roofline": {"bound": "hbm" if achieved <= HBM_PEAK_GBPS else "cache", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS if achieved <= HBM_PEAK_GBPS else None,
                         **({"algorithmic_over_hbm_peak": achieved / HBM_PEAK_GBPS,
                             "bound_note": "the looked-up rows are served from the caches (ids concentrated on rows that stay resident): the algorithmic byte "
                                           "rate exceeds the HBM peak, so no HBM-roofline fraction is given; the DRAM traffic of this id distribution was not "
                                           "profiled (traffic: null)"} if achieved > HBM_PEAK_GBPS else {}),
                         "traffic": traffic, "traffic_unit": "bytes/launch",
                         **({"frac_with_cross_write": achieved_cw / HBM_PEAK_GBPS, "achieved_with_cross_write": achieved_cw}
                            if survey_bytes != bytes_per_impr else {}),
                         "kernel_ms_mean": kern_ms, "algorithmic_bytes_per_launch": survey_bytes * BATCH,
                         "granule_bytes_per_launch": None if granule is None else granule * BATCH,
                         "stream_copy_GBps": stream_copy,
                         "traffic_GBps": None if traffic is None else traffic / (kern_ms * 1e-3) / 1e9,
                         "traffic_frac_of_stream_copy": None if (traffic is None or not stream_copy) else traffic / (kern_ms * 1e-3) / 1e9 / stream_copy,
                         "ceiling_note": "granule_bytes_per_launch = the same lookups at the memory system's granules (random row reads in whole 128-byte "
                                         "lines, writes in 64-byte requests); stream_copy_GBps = a plain 1 GiB copy (read + written bytes / time) measured in "
                                         "this run on this box.  frac can not exceed (algorithmic / granule bytes) x (stream copy / peak); "
                                         "traffic_frac_of_stream_copy says how close the launch's REAL traffic rate is to the box's copy rate",
                         "note": "achieved = algorithmic bytes per launch / mean launch duration (HIP events around "
                                 "the timed region on the launch stream / steps); traffic = DRAM bytes per launch from "
                                 "the committed rocprofv3 PMC passes (profiles/traffic.json), null if not profiled"},
        }
        if distinct is not None:
            out["distinct_output_buffers"] = distinct
        if wide_split is not None:
            out[wide_split["leg"]] = wide_split
        if fwd_bwd is not None:
            out["fwd_bwd"] = fwd_bwd
        if module_path is not None:
            out["module_path"] = module_path
        if dcn_v2_cross is not None:
            out["dcn_v2_cross"] = dcn_v2_cross
        if input_pipeline is not None:
            out["input_pipeline"] = input_pipeline
        if world == 1 and not args.no_cpu_baseline and not args.force_sharded:
            big = sum(t.numel() for t in path.tables) * 4 > (64 << 30)      # host copies of > 64 GB of tables: skip
            if not big:
                out["cpu_baseline"] = cpu_baseline(path)
                out["cpu_baseline_torch"] = cpu_baseline_torch(path)
            else:
                out["cpu_baseline"] = None
        try:
            info = ops.device_info(local_rank)
            out["config"]["device"] = {"compute_units": info["compute_units"], "clock_khz": info["clock_khz"],
                                       "hbm_gib": round(info["global_mem_bytes"] / 2 ** 30, 1)}
        except Exception:
            pass
        if hasattr(path, "n_sharded"):
            out["config"]["layout"] = {"mode": args.shard_mode, "tables_row_sharded": path.n_sharded, "tables_replicated": path.n_replicated,
                                       "engine": getattr(path, "engine", "legacy"),
                                       "levers": "SURVEY 7 hard-part 1: (a) replication of small tables -- OFF in the row headline (that is `other_layout`), "
                                                 "(b) per-destination dedup -- off (uniform ids have nothing to dedup; RowShardedEmbedding(dedup=True) exists), "
                                                 "(c) owner-side pooling of bag features -- on (c4), (d) int32 owner ids on the wire, fixed-capacity "
                                                 "equal-split all-to-alls, no host read inside a step -- on"}
        if world > 1:
            # what the process group itself reports (not the launcher's environment): the N > 1 line must be able to show that N ranks exchanged
            out["rccl_ranks"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                 "transport": "host-staged gloo (NRX_BENCH_HOST_STAGED test transport: NOT a measurement)" if staged else "RCCL over xGMI"}
            ref = _direct_1gpu_reference(args.workload)
            if ref is not None:
                out["scaling_vs_1gpu"] = {"direct_1gpu_value": ref["value"], "direct_1gpu_source": ref["source"],
                                          "ratio": out["value"] / ref["value"], "per_gpu_efficiency": out["value"] / world / ref["value"],
                                          "note": "this line's whole-job value over the DIRECT single-GPU path's committed number of the same workload (no "
                                                  "exchange at all): the price of the row-sharded exchange is inside the ratio.  The driver computes scaling "
                                                  "efficiency itself from the per-N lines; this is the self-description"}
        return out
    base_line = _build_line() if rank == 0 else None      # everything of the headline, before any secondary leg runs

    def _print_line(note):
        out = base_line
        if planner is not None:
            out["other_layout"] = planner
        if strong is not None:
            out["strong_scaling"] = strong
        if a2a is not None:
            out["a2a"] = a2a
        if sharded_fb is not None:
            out["fwd_bwd"] = sharded_fb
        if note or secondary_note:
            out["secondary_note"] = note or secondary_note
        line = json.dumps(out)
        print(line, flush=True)
        if os.environ.get("NRX_BENCH_OUT") and os.environ.get("NRX_BENCH_CHILD") != "1":      # under the self-launcher only its merged line is filed
            with open(os.environ["NRX_BENCH_OUT"], "a") as f:
                f.write(line + "\n")


    def emit(note=None):
        if emitted["done"]:
            return
        emitted["done"] = True
        if rank == 0:
            _print_line(note)

    if world > 1 and not args.headline_only:
        # secondary legs (never the headline `value`).  None of the multi-rank collectives below could be exercised on real
        # xGMI while this was written: a watchdog prints the headline line and leaves if they stall
        import threading

        def on_timeout():
            # the headline line above is complete and valid (its timed region ended before any secondary leg started); the
            # process still leaves NON-ZERO: a collective stalled, and a hung multi-rank run must not report success
            emit(f"secondary legs did not finish within {args.secondary_timeout:.0f} s: headline only; exit code 3")
            os._exit(3)

        dog = threading.Timer(args.secondary_timeout, on_timeout)
        dog.daemon = True
        dog.start()
        from news_recsys_amd.sharding import ShardedBenchPath
        try:
            probe = path.a2a_probe(min(args.steps, 50)) if hasattr(path, "a2a_probe") else None
            if probe is not None:
                t = torch.tensor([probe["ms"]], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = t.item()
                remote = probe["bytes"] * (world - 1) / world
                a2a = {"bytes_per_rank_per_step": probe["bytes"], "ms": ms, "GBps_per_rank": remote / (ms * 1e-3) / 1e9,
                       "GBps_per_link": remote / (world - 1) / (ms * 1e-3) / 1e9,
                       "xgmi_link_peak_GBps": 153.0, "xgmi_links_per_gpu": 7,
                       "frac_of_link_peak": remote / (world - 1) / (ms * 1e-3) / 1e9 / 153.0,
                       "what": probe.get("what"),
                       "note": "equal splits, max over ranks; per link = remote bytes / (N-1) point-to-point xGMI links (7 x ~153 GB/s per GPU)"}
            if hasattr(path, "train_setup"):
                sharded_fb = sharded_train_leg(path)
            del path
            torch.cuda.empty_cache()
            bs = BATCH // world
            p3 = ShardedBenchPath(args.workload, device, seed, rank, world, bs, args.shard_mode, host_staged=staged)
            dt3, _ = timed(p3.step)
            if not p3.overflowed():
                strong = {"value": bs * world * args.steps / dt3, "unit": "impressions/s", "ms_per_step": dt3 * 1e3 / args.steps,
                          "global_batch": bs * world, "batch_per_gpu": bs, "layout_mode": args.shard_mode,
                          "note": "strong scaling: the global batch stays 65536, every rank takes 65536 / N of it"}
            del p3
            torch.cuda.empty_cache()
            other = "row" if args.shard_mode == "auto" else "auto"
            p2 = ShardedBenchPath(args.workload, device, seed, rank, world, BATCH, other, host_staged=staged)
            dt2, _ = timed(p2.step)
            if not p2.overflowed():
                planner = {"layout_mode": other, "value": BATCH * world * args.steps / dt2, "unit": "impressions/s",
                           "ms_per_step": dt2 * 1e3 / args.steps, "layout": p2.desc, "tables_replicated": p2.n_replicated,
                           "tables_row_sharded": p2.n_sharded,
                           "note": "the other table layout, same workload and steps; xGMI is point-to-point (one ~153 GB/s link "
                                   "per GPU pair), so an all-row-sharded row return is per-link bound, worst at N=2"}
            del p2
        except Exception as e:          # noqa: BLE001 -- a failed secondary leg must not cost the headline
            secondary_note = f"a secondary leg failed: {type(e).__name__}: {e}"
        dog.cancel()

    emit()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
