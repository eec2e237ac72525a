#!/usr/bin/env python3
"""Dev (GPU box): tests/stress_shard_step.py's random towers at WORLD 2 / 3 -- rank processes sharing cuda:0, exchanging through gloo with host-staged
buffers (the transport of tests/test_shard_step_multirank_one_gpu.py: RCCL refuses two ranks on one device), every launch the product's.  All ranks
draw the same configurations from one seed (tables, every rank's batch, the form of the step) and stay in lockstep; each rank checks ITS part against a
float64 restatement on the rank-major concatenation of the batches (F.embedding + the pooling of src/model/BaseModel/base_model.py:262-282, autograd):
  * its rows of the forward concat: single-valued columns bit for bit, pooled columns (and the FM logit) within the fp32 summation tolerance;
  * its shard of every table's gradient -- the sum of its (key, value) lists on the rows it owns (global row = (arena row - 1) x world + rank) --
    within tolerance of the concatenated batch's gradient on those rows; nothing on the padding row;
  * two runs word for word equal.
Forms drawn per tower: one-sided placement / the all-to-all forward, the requester's pack as the owner's placement pass / the buffered backward,
NRX_ROUTE_BAGS = runs | one | legacy, binary-mask fast path, exchange groups side by side or not; one tower in four is an FM plan.
usage: python tests/stress_shard_step_multirank.py [seconds=120] [seed=1] [world=2]"""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = "cuda:0"


def _worker(rank, world, port, budget, seed, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import faulthandler, warnings
    faulthandler.enable()
    faulthandler.dump_traceback_later(float(os.environ.get("NRX_STRESS_DUMP_AFTER", budget + 60)), exit=True)      # a rank stuck in a collective says where
    warnings.simplefilter("ignore")
    from tests import _poison
    from news_recsys_amd import shard_step
    from news_recsys_amd._lib import NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM, NRX_SPARSE
    from news_recsys_amd.sharding import RowShardedEmbedding, ShardedFeature
    rng = np.random.default_rng(seed)                   # the SAME stream on every rank
    t0, n_done, n_iter = time.time(), 0, 0
    only = int(os.environ.get("NRX_STRESS_ONLY", "0"))
    try:
        while True:
            go = torch.tensor([1 if time.time() - t0 < budget else 0])
            dist.broadcast(go, 0)
            if not int(go.item()):
                break
            _poison.poison()
            B = int(rng.choice([1, 3, 64, 81, 700, 2500, 6000]))
            idt = torch.int64 if rng.integers(0, 3) else torch.int32
            fm = bool(rng.integers(0, 4) == 0)
            dims = sorted(set(int(d) for d in rng.choice([16, 32, 64], 1 if fm else int(rng.integers(1, 3)))))
            tables, feats, ids_all, ws_all = {}, [], [], []
            for d in dims:
                for t in range(int(rng.integers(1, 3))):
                    tables[f"t{d}_{t}"] = (int(rng.choice([2, 50, 3000, 200000])), d)
            names = list(tables)
            look = 0
            for f in range(int(rng.integers(1, 9))):
                t = names[int(rng.integers(0, len(names)))]
                rows, d = tables[t]
                skew = rng.integers(0, 3) == 0
                x = rng.integers(0, rows, (world, B)) if not skew else np.minimum(rng.zipf(1.3, (world, B)) - 1, rows - 1)
                feats.append(ShardedFeature(f"s{f}", NRX_SPARSE, t, d, 0, False, fm))
                ids_all.append(np.asarray(x, np.int64))
                ws_all.append(None)
                look += B
            binary_ok = True
            for d in dims:
                if fm or rng.integers(0, 2) == 0:
                    continue
                bag_table = [n for n in names if tables[n][1] == d][0]
                rows = tables[bag_table][0]
                for f in range(int(rng.integers(1, 3))):
                    kind = int(rng.choice([NRX_BAG_MASKED_MEAN, NRX_BAG_MEAN, NRX_BAG_SUM]))
                    L = int(rng.choice([1, 2, 4, 17, 50, 81]))
                    x = rng.integers(0, rows, (world, B, L))
                    w = None
                    if kind != NRX_BAG_MEAN:
                        m = (np.arange(L)[None, None, :] < rng.integers(0, L + 1, (world, B, 1))).astype(np.float32)
                        x = x * m.astype(np.int64)
                        if kind == NRX_BAG_SUM and rng.integers(0, 2):
                            m = m * rng.random((world, B, L)).astype(np.float32)
                            binary_ok = False
                        w = m
                    feats.append(ShardedFeature(f"b{d}_{f}", kind, bag_table, d, L))
                    ids_all.append(np.asarray(x, np.int64))
                    ws_all.append(w)
                    look += B * L
            forms = (bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), str(rng.choice(["runs", "one", "legacy"])), bool(rng.integers(0, 2)), str(rng.choice(["1", "0"])))
            width = sum(f.dim for f in feats)
            full = {n: rng.standard_normal((r, d)).astype(np.float32) for n, (r, d) in tables.items()}
            for x in full.values():
                x[0] = 0
            ups = rng.standard_normal((world, B, width)).astype(np.float32)
            ups_fm = rng.standard_normal((world, B)).astype(np.float32)
            if look > 400_000:
                continue
            n_iter += 1
            if only and n_iter != only:                                  # (debugging: NRX_STRESS_ONLY=<k> runs the k-th tower alone)
                continue
            one_sided, direct, route, binary, overlap = forms
            route = os.environ.get("NRX_STRESS_ROUTE", route)            # (debugging: pin a form)
            overlap = os.environ.get("NRX_STRESS_OVERLAP", overlap)
            one_sided = bool(int(os.environ.get("NRX_STRESS_ONE_SIDED", int(one_sided))))
            direct = bool(int(os.environ.get("NRX_STRESS_DIRECT", int(direct))))
            binary = bool(int(os.environ.get("NRX_STRESS_BINARY", int(binary))))
            binary = binary and binary_ok
            if os.environ.get("NRX_STRESS_VERBOSE"):
                print(f"[rank {rank}] tower {n_done + 1} begins: B={B} forms={forms} fm={fm} idt={idt} tables={tables} "
                      f"feats={[(f.name, f.kind, f.table, f.dim, f.bag_len) for f in feats]}", flush=True)
            os.environ["NRX_ROUTE_BAGS"], os.environ["NRX_SHARD_OVERLAP"] = route, overlap
            # ---- float64 restatement on the concatenated batch
            t64 = {n: torch.from_numpy(x).to(DEV).double().requires_grad_() for n, x in full.items()}
            outs, cols, col = [], [], 0
            for f, x, w in zip(feats, ids_all, ws_all):
                xc = torch.from_numpy(x.reshape((world * B,) + x.shape[2:])).to(DEV)
                e = torch.nn.functional.embedding(xc, t64[f.table])
                if f.kind == NRX_SPARSE:
                    outs.append(e)
                elif f.kind == NRX_BAG_MASKED_MEAN:
                    wd = torch.from_numpy(w.reshape(world * B, -1)).to(DEV).double()
                    outs.append((e * wd.unsqueeze(-1)).sum(1) / (wd.sum(1, keepdim=True) + 1e-8))
                elif f.kind == NRX_BAG_MEAN:
                    outs.append(e.mean(1))
                else:
                    outs.append((e * torch.from_numpy(w.reshape(world * B, -1)).to(DEV).double().unsqueeze(-1)).sum(1))
                cols.append((col, f.dim, f.kind))
                col += f.dim
            ref_out = torch.cat(outs, 1)
            up_cat = torch.from_numpy(ups.reshape(world * B, width)).to(DEV).double()
            loss = (ref_out * up_cat).sum()
            ref_fm = None
            if fm:
                e3 = torch.stack(outs, 1)
                v = e3[:, :, 1:]
                ref_fm = e3[:, :, 0].sum(1) + 0.5 * ((v.sum(1) ** 2) - (v ** 2).sum(1)).sum(1)
                loss = loss + (ref_fm * torch.from_numpy(ups_fm.reshape(-1)).to(DEV).double()).sum()
            ref_g = dict(zip(t64, torch.autograd.grad(loss, list(t64.values()), allow_unused=True)))
            n_max = {n: 1 for n in tables}
            for f, x in zip(feats, ids_all):
                vv = torch.from_numpy(x.reshape(-1))
                c = torch.bincount(vv[vv > 0], minlength=1)
                n_max[f.table] += int(c.max().item()) if c.numel() else 0
            # ---- this rank's step
            arenas = {n: shard_step.make_arena(x.shape[0], x.shape[1], rank, world, DEV, full=torch.from_numpy(x).to(DEV)) for n, x in full.items()}
            ins = [torch.from_numpy(x[rank]).to(DEV).to(idt) for x in ids_all]
            ws = [None if w is None else torch.from_numpy(w[rank]).to(DEV) for w in ws_all]
            up = torch.from_numpy(ups[rank]).to(DEV)
            up_fm = torch.from_numpy(ups_fm[rank]).to(DEV) if fm else None
            eng = RowShardedEmbedding(rank, world, slack=3.0, host_staged=True, overflow_policy="defer")
            step = shard_step.PreparedShardedStep(eng, feats, ins, ws, arenas, one_sided=one_sided, binary_masks=binary, check_index=True)
            step.bind_backward(up, up_fm, direct_grad=direct)
            what = dict(rank=rank, world=world, B=B, feats=[(f.name, f.kind, f.table, f.dim, f.bag_len) for f in feats], tables=tables, forms=forms, idt=str(idt), fm=fm)
            runs = []
            for _ in range(2):
                out, _, fmv = step.run()
                entries = step.backward()
                torch.cuda.synchronize()
                dist.barrier()                      # (one-sided forms: every peer's launches into this rank's buffers have finished)
                runs.append((out.clone(), None if fmv is None else fmv.clone(),
                             [(e["uniq"][:int(e["counts"][0])].clone(), e["values"][:int(e["counts"][0])].clone()) for e in entries]))
                dist.barrier()                      # (nobody starts the next run while a peer still reads this one)
            step.check()
            assert torch.equal(runs[0][0].view(torch.int32), runs[1][0].view(torch.int32)), ("two runs, forward", what)
            for (k0, v0), (k1, v1) in zip(runs[0][2], runs[1][2]):
                assert torch.equal(k0, k1) and torch.equal(v0.view(torch.int32), v1.view(torch.int32)), ("two runs, gradient", what)
            out = runs[0][0]
            mine = ref_out[rank * B:(rank + 1) * B].detach()
            for c0, d, kind in cols:
                got, want = out[:, c0:c0 + d], mine[:, c0:c0 + d]
                if kind == NRX_SPARSE:
                    assert torch.equal(got, want.float()), ("single-valued columns", what)
                else:
                    assert (got.double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), ("pooled columns", what)
            if fm:
                rf = ref_fm[rank * B:(rank + 1) * B].detach()
                assert (runs[0][1].double() - rf).abs().max().item() <= 2e-5 * max(1.0, rf.abs().max().item()) + 1e-5 * len(feats), ("FM logit", what)
            got_g = {n: torch.zeros((-(-(t.shape[0] - rank) // world) if t.shape[0] > rank else 0, t.shape[1]), dtype=torch.float64, device=DEV) for n, t in t64.items()}
            for e, (keys, vals) in zip(entries, runs[0][2]):
                for ti, arena in enumerate(e["tables"]):
                    name = next(n for n, a in arenas.items() if a is arena)
                    sel = (keys >> 40) == ti
                    rows = keys[sel] & ((1 << 40) - 1)
                    live = rows > 0
                    got_g[name].index_add_(0, rows[live] - 1, vals[sel][live].double())
            for n in tables:
                r = ref_g[n] if ref_g[n] is not None else torch.zeros_like(t64[n])
                r = r.clone()
                r[0] = 0
                want = r[rank::world]
                assert got_g[n].shape == want.shape, (n, got_g[n].shape, want.shape, what)
                if rank == 0 and want.shape[0]:
                    assert got_g[n][0].abs().max().item() == 0, ("padding row", what)
                err = (got_g[n] - want).abs().max().item() if want.numel() else 0.0
                rmax = max(1.0, r.abs().max().item())
                assert err <= 2e-5 * rmax + 50 * 6e-8 * rmax * n_max[n] ** 0.5, dict(table=n, err=err, n_max=n_max[n], rmax=rmax, **what)
            n_done += 1
            if os.environ.get("NRX_STRESS_VERBOSE"):
                print(f"[rank {rank}] tower {n_done}: B={B} feats={len(feats)} forms={forms} fm={fm} idt={idt}", flush=True)
            del step
            dist.barrier()
        q.put((rank, n_done, None))
    except Exception as e:          # noqa: BLE001
        import traceback
        msg = traceback.format_exc()[-3000:]
        print(f"[rank {rank}] FAILED after {n_done} towers:\n{msg}", flush=True)
        q.put((rank, n_done, msg))
        q.close()
        q.join_thread()             # (the message is with the parent before this process goes)
        os._exit(1)                 # (the peers are stuck in a collective: the parent kills them)
    finally:
        try:
            dist.destroy_process_group()
        except Exception:          # noqa: BLE001
            pass


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    from tests.test_sharding_gloo import _free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, budget, seed, q)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    res, bad = {}, None
    while len(res) < world and bad is None:
        try:
            rank, n, err = q.get(timeout=budget + 150)
        except Exception:          # noqa: BLE001
            bad = "timeout waiting for the ranks"
            break
        res[rank] = n
        if err is not None:
            bad = f"rank {rank} after {n} towers:\n{err}"
    for p in procs:
        p.join(timeout=5 if bad else 120)
        if p.is_alive():
            p.kill()
    if bad:
        print("stress_shard_step_multirank: FAILED\n" + bad)
        sys.exit(1)
    print(f"stress_shard_step_multirank: world {world}, {min(res.values())} random towers in lockstep: every rank's forward rows and gradient shard within tolerance of "
          f"float64 on the concatenated batch (single-valued columns bit for bit), two runs word for word equal ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
