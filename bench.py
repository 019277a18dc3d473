#!/usr/bin/env python
"""bench.py — the headline metric of BASELINE.json: PAF CIGAR-ops/s for `paf2maf`+`stat`.

One "step" = one pass of the hot path over one HBM-resident batch of synthetic PAF records:
  K1 wga_cigar_stat  (parse_paf_to_cigar, cigar.rs:629-707)  -> per-record counts
  wga_paf2maf_layout (row geometry, converter.rs:237-262)
  K2 wga_paf2maf_expand (parse_cigar_to_insert + reverse_complement, cigar.rs:492-551)
  + a reduction of the stat totals (RCCL all-reduce over xGMI when N > 1)
Workload at N=1 = BASELINE.json configs[1]: 100 000 records, mean CIGAR 5 kop, 2 x 50 Mb
sequence pools.  N > 1: weak scaling — every rank owns its own shard of records (records are
independent; real runs shard by hash(target_name)), no data-path collective.

Prints ONE JSON line on rank 0.  `python bench.py` defaults to N=1, 5 steps, 2 warm-up steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy rate


def cpu_baseline(tb, budget_s=12.0, max_records=40000):
    """The oracle (oracle/oracle.c: reference-faithful port, text tokenising + String::insert_str
    tail memmoves) timed on the host over a bounded sample of the same batch.  `value` is ONE core —
    the reference's paf2maf is a serial loop (converter.rs:196) — and `all_cores` shows the same
    per-record work spread over every host core (what a rayon-parallel paf2maf would get; the
    reference only parallelises stat)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_py as orc
    from concurrent.futures import ThreadPoolExecutor
    n = min(tb["n"], max_records)
    op_off = tb["op_off"][: n + 1].cpu().numpy()
    ops = tb["ops"][: int(op_off[n])].cpu().numpy().view(np.uint32)
    t_pool, q_pool = tb["t_pool"].cpu().numpy(), tb["q_pool"].cpu().numpy()
    to, tl = tb["t_src_off"][:n].cpu().numpy(), tb["t_src_len"][:n].cpu().numpy()
    qo, ql = tb["q_src_off"][:n].cpu().numpy(), tb["q_src_len"][:n].cpu().numpy()
    strand = tb["strand_neg"][:n].cpu().numpy()
    orc.lib()

    def prep(i):                                    # input preparation: never timed
        a, b = int(op_off[i]), int(op_off[i + 1])
        return (orc.ops_to_text(ops[a:b]), t_pool[int(to[i]):int(to[i] + tl[i])].tobytes(),
                q_pool[int(qo[i]):int(qo[i] + ql[i])].tobytes(), int(strand[i]), b - a)

    def work(x):
        cg, t, q, neg, nops = x
        t0 = time.perf_counter()
        orc.parse_paf_to_cigar(cg, neg)                       # stat
        if neg:
            q = orc.reverse_complement(q)                     # paf2maf
        orc.parse_cigar_to_insert(cg, t, q)
        return nops, time.perf_counter() - t0

    ops_done, done, t_work = 0, 0, 0.0
    t_start = time.perf_counter()
    for i in range(n):
        o, dt = work(prep(i))
        ops_done += o
        t_work += dt
        done += 1
        if t_work > budget_s or time.perf_counter() - t_start > 4 * budget_s:
            break
    res = {"value": ops_done / t_work, "unit": "ops/s", "cores": 1, "kind": "port",
           "sample": "first %d records (%d ops) of the same batch, stat + paf2maf per record, "
                     "%.1f s of oracle time on 1 core" % (done, ops_done, t_work)}
    # the same per-record work over all cores (ctypes drops the GIL inside the oracle calls)
    cores = os.cpu_count() or 1
    if cores > 1:
        m = min(n, 8000)
        items = [prep(i) for i in range(m)]
        t1 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            tot = sum(o for o, _ in ex.map(work, items, chunksize=4))
        wall = time.perf_counter() - t1
        res["all_cores"] = {"value": tot / wall, "unit": "ops/s", "cores": cores,
                            "sample": "%d records (%d ops) in %.2f s wall over %d threads" % (m, tot, wall, cores)}
    return res


def pmc_traffic(args, job):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE and
    WRITE_SIZE are collected in their own rocprofv3 runs — scripts/gpu_pmc.sh — and corrected as
    MI355X_MICROARCH.md prescribes; they cannot be read live).  Only valid for the workload they
    were measured on."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None
    w = t.get("workload", {})
    if w.get("records") != args.records or w.get("mean_ops") != args.mean_ops or w.get("ops") != job.n_ops:
        return None
    return t["hbm_bytes_per_launch"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--records", type=int, default=100_000)
    ap.add_argument("--mean-ops", type=int, default=5000)
    ap.add_argument("--pool-mb", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", type=int, default=8, help="records spot-checked against the oracle")
    ap.add_argument("--param", action="append", default=[], help="engine test knob name=value")
    ap.add_argument("--neg-frac", type=float, default=0.5)
    ap.add_argument("--m-only", action="store_true", help="variant of configs[1] with = / X merged into M ops")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from wgatools_amd import engine, pipeline, synth
    eng = engine.Engine(local_rank)
    for kv in args.param:
        k, v = kv.split("=")
        eng.set_param(k, int(v))
    seed = 0x5747415F + 2 + rank
    tb = synth.make_paf_batch_torch(seed, args.records, args.mean_ops, args.pool_mb * 1_000_000, dev,
                                    neg_frac=args.neg_frac, use_m=args.m_only)
    job = pipeline.Paf2MafStatJob(eng, tb)
    job.bind_stream()
    totals = torch.zeros(11, dtype=torch.int64, device=dev)

    def step(evs=None):
        if evs:
            evs[0].record()
        job.stat()
        if evs:
            evs[1].record()
        job.layout()
        if evs:
            evs[2].record()
        job.expand()
        if evs:
            evs[3].record()
        eng.counts_total(job.n, job.counts, totals)          # global stat totals (88 bytes)
        if world > 1:
            dist.all_reduce(totals)                          # RCCL over xGMI: 88 bytes

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events inside the library bracket the dominant kernel alone (on the launch stream)
    eng.set_param("expand_timing", 1)
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    nops = torch.tensor([float(job.n_ops)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(nops)
    elapsed = float(el.item())
    total_ops = float(nops.item())

    # the run only counts if every record came out clean
    if not args.param:
        assert bool((job.diag == -1).all()), "kernel reported per-record errors on clean synthetic input"

    if rank == 0:
        k_stat = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
        k_layout = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps
        k_expand_call = sum(e[2].elapsed_time(e[3]) for e in events) / args.steps   # pre-pass + kernel
        ms_sum, n_timed = eng.expand_timing()
        k_expand = ms_sum / n_timed if n_timed else k_expand_call
        ab = job.algorithmic_bytes()
        in_bytes = 4 * job.n_ops + int(tb["t_src_len"].sum()) + int(tb["q_src_len"].sum())
        ach = ab["expand"] / (k_expand * 1e-3) / 1e9
        result = {
            "metric": "paf_cigar_ops_per_s (paf2maf+stat)",
            "value": total_ops * args.steps / elapsed,
            "unit": "ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: paf2maf+stat, %d records/GPU x mean %d ops "
                            "(lognormal s=0.5; =/X/I/D mix), 2 x %d Mb pools, strand 50/50, "
                            "HBM-resident" % (args.records, args.mean_ops, args.pool_mb),
                "records_per_gpu": args.records, "ops_per_gpu": job.n_ops,
                "columns_per_gpu": int((tb["mx"] + tb["i"] + tb["d"]).sum()),
                "output_bytes_per_gpu": job.out_bytes, "sharding": "records, no data-path collective",
            },
            "input_GBps": in_bytes * world * args.steps / elapsed / 1e9,
            "kernel_ms": {"k_cigar_stat": k_stat, "layout_scan": k_layout, "k_paf2maf_expand": k_expand,
                          "expand_prepass (k_rec_desc + k_tile_base)": k_expand_call - k_expand},
            "roofline": {
                "kernel": "k_paf2maf_expand", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(args, job),
                "algorithmic_bytes_per_launch": ab["expand"],
                "k_cigar_stat_GBps": ab["stat"] / (k_stat * 1e-3) / 1e9,
            },
        }
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only
            result["cpu_baseline"] = cpu_baseline(tb)
            # part of the same leg (the only place bench.py touches oracle/): a few of the rows the timed steps
            # wrote, compared with the oracle's rows for those records
            if args.check:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import parity_cases as pc
                step_idx = torch.linspace(0, tb["n"] - 1, args.check).long().tolist()
                for i in step_idx:
                    r = synth.torch_batch_record_to_numpy(tb, i)
                    et, eq = pc.oracle_rows(r, 0)
                    gt, gq = job.record_rows(i)
                    assert gt == et and gq == eq, "record %d differs from the oracle" % i
                result["cpu_baseline"]["parity_spot_check"] = "%d records bit-identical to oracle rows" % len(step_idx)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
