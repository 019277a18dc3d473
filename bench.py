#!/usr/bin/env python
"""bench.py — the headline metric of BASELINE.json: PAF CIGAR-ops/s for `paf2maf`+`stat`.

One "step" = one pass of the hot path over one HBM-resident batch of synthetic PAF records:
  K1 wga_cigar_stat  (parse_paf_to_cigar, cigar.rs:629-707)  -> per-record counts
  wga_paf2maf_layout (row geometry, converter.rs:237-262)
  K2 wga_paf2maf_expand (parse_cigar_to_insert + reverse_complement, cigar.rs:492-551)
  + a reduction of the stat totals (RCCL all-reduce over xGMI when N > 1)
Workload at N=1 = BASELINE.json configs[1]: 100 000 records, mean CIGAR 5 kop, 2 x 50 Mb
sequence pools.  N > 1: the SAME global batch (strong scaling: total work fixed), every rank takes the records
fnv1a64(target_name) % N gives it, one all-reduce of the per-record output sizes per step (ordered output) and one of the
stat totals; row bytes never cross GPUs.  `--scaling weak` gives every rank its own shard of --records instead.

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


def host_info():
    """CPU model / sockets / cores / threads / NUMA nodes of the box the baseline runs on, and how the port was built"""
    info = {}
    try:
        import subprocess
        for line in subprocess.run(["lscpu"], stdout=subprocess.PIPE, text=True).stdout.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "NUMA node(s)", "CPU(s)"):
                info[k] = v
    except Exception:
        pass
    info["usable_threads"] = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    info["numa_policy"] = "none (threads float; first-touch)"
    info["build"] = "gcc -O2 -std=gnu11, pthreads (oracle/Makefile)"
    return info


def cpu_baseline(tb, budget_s=8.0, max_records=40000):
    """oracle/cpu_bench.c on the host cores over a bounded sample (the first records) of the same batch, BASELINE.md
    section 3's three forms:
      ref_faithful_1    stat + paf2maf exactly as the reference structures them (text tokenised per consumer,
                        String::insert_str tail memmoves), one thread — the reference's default -t 1
      ref_faithful_all  the same with -t <all cores>: stat spreads over the cores (rayon par_bridge, stat.rs:67-81),
                        paf2maf stays the serial loop it is in the reference (converter.rs:196)
      optimised_all     one pass over packed ops, linear-time rows, every core on both tools
    `value` = ref_faithful_1 (cores 1).  Baselines, not targets: the GPU / CPU ratio says nothing about kernel quality."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_py as orc
    orc.lib()
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_all = min(tb["n"], max_records)
    op_off = tb["op_off"][: n_all + 1].cpu().numpy().astype(np.uint64)
    ops = tb["ops"][: int(op_off[n_all])].cpu().numpy().view(np.uint32)
    t_pool, q_pool = tb["t_pool"].cpu().numpy(), tb["q_pool"].cpu().numpy()

    def arrays(n, with_text=True):
        blob = cg_off = None
        if with_text:   # input preparation: never timed
            texts = [orc.ops_to_text(ops[int(op_off[i]):int(op_off[i + 1])]) for i in range(n)]
            texts = [t if isinstance(t, bytes) else t.encode() for t in texts]
            cg_off = np.zeros(n + 1, dtype=np.uint64)
            cg_off[1:] = np.cumsum([len(t) for t in texts])
            blob = np.frombuffer(b"".join(texts), dtype=np.uint8).copy()
        f = lambda k: np.ascontiguousarray(tb[k][:n].cpu().numpy().astype(np.uint64))
        return (blob, cg_off, ops, np.ascontiguousarray(op_off[: n + 1]), np.ascontiguousarray(tb["strand_neg"][:n].cpu().numpy()),
                t_pool, f("t_src_off"), f("t_src_len"), q_pool, f("q_src_off"), f("q_src_len"))

    # pilot: how many records fit the budget of the slowest leg (serial quadratic paf2maf)
    pilot = min(n_all, 200)
    r = orc.bench_run(1, 1, *arrays(pilot))
    per_rec = max(r.seconds / pilot, 1e-6)
    n = int(max(pilot, min(n_all, budget_s / per_rec)))
    a = arrays(n)
    stat1 = orc.bench_run(0, 1, *a)
    p2m1 = orc.bench_run(1, 1, *a)
    stat_all = orc.bench_run(0, threads, *a)
    p2m_all = orc.bench_run(1, threads, *a)
    big = arrays(n_all, with_text=False)   # the packed-op port is fast: it gets the larger sample
    opt1 = orc.bench_run(2, 1, *big)
    opt_all = orc.bench_run(2, threads, *big)
    nops = int(stat1.ops)
    rate = lambda secs: nops / secs
    sample = "first %d records (%d ops) of the same batch" % (n, nops)
    return {
        "value": rate(stat1.seconds + p2m1.seconds), "unit": "ops/s", "cores": 1, "kind": "port",
        "sample": sample + ": stat %.2f s + paf2maf %.2f s on one core" % (stat1.seconds, p2m1.seconds),
        "ref_faithful_1": {"value": rate(stat1.seconds + p2m1.seconds), "unit": "ops/s", "cores": 1,
                           "stat_ops_per_s": rate(stat1.seconds), "paf2maf_ops_per_s": rate(p2m1.seconds)},
        "ref_faithful_all": {"value": rate(stat_all.seconds + p2m1.seconds), "unit": "ops/s", "cores": threads,
                             "stat_ops_per_s": rate(stat_all.seconds), "paf2maf_ops_per_s": rate(p2m1.seconds),
                             "note": "stat over %d threads (par_bridge); paf2maf is a serial loop in the reference "
                                     "(converter.rs:196) and stays on one core" % threads,
                             "paf2maf_if_it_were_parallel_ops_per_s": rate(p2m_all.seconds)},
        "optimised_all": {"value": opt_all.ops / opt_all.seconds, "unit": "ops/s", "cores": threads,
                          "one_core_ops_per_s": opt1.ops / opt1.seconds,
                          "sample": "first %d records (%d ops)" % (n_all, int(opt1.ops)),
                          "note": "single pass over packed ops, linear-time rows, stat + paf2maf fused"},
        "host": host_info(),
    }


def kernel_source_sha():
    """the row kernel's source: roofline.traffic is only valid for the kernel it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("wga_kernels.h", "wga_kernels_k2s.h", "wga_intrin.h"):
        h.update(open(os.path.join(ROOT, "wgatools_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(args, job):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in their own rocprofv3 runs — scripts/gpu_pmc.sh — and corrected as MI355X_MICROARCH.md prescribes; they
    cannot be read live).  Only valid for the workload AND the kernel source they were measured on: anything else -> null."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True):  # newest round first
        try:
            t = json.load(open(path))
        except Exception:
            continue
        w = t.get("workload", {})
        if w.get("records") != args.records or w.get("mean_ops") != args.mean_ops or w.get("ops") != job.n_ops:
            continue
        if t.get("kernel_source_sha") != kernel_source_sha():
            continue
        return t["hbm_bytes_per_launch"]
    return None


def extra_shape(eng, synth, pipeline, torch, dev, seed, records, mean_ops, pool_mb, steps=3):
    """K2 alone on another shape of the workload (the pool size and the record length move the fraction: the default
    50 Mb pools sit in the 256 MB Infinity Cache) -> (kernel ms, fraction of the HBM peak)"""
    tb = synth.make_paf_batch_torch(seed, records, mean_ops, pool_mb * 1_000_000, dev)
    job = pipeline.Paf2MafStatJob(eng, tb)
    job.bind_stream()
    for _ in range(3):    # first touch of the output buffer
        job.step()
    torch.cuda.synchronize()
    eng.expand_timing()   # drop what the warm-up recorded
    for _ in range(steps):
        job.step()
    torch.cuda.synchronize()
    ms_sum, n_timed = eng.expand_timing()
    ms = ms_sum / max(1, n_timed)
    ok = bool((job.diag == -1).all())
    frac = job.algorithmic_bytes()["expand"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    del job, tb
    torch.cuda.empty_cache()
    assert ok, "kernel reported per-record errors on clean synthetic input"
    return ms, frac


def allocation_spread(eng, pipeline, torch, tb, first_ms, n_alloc=4, steps=3, warm=2):
    """The row kernel on further FRESH output allocations of this process, all held at once (so they are different memory):
    the kernel's time depends on where the 15 GB of rows land (+- 5 % from box to box and from allocation to allocation:
    VERDICT r05), so the line carries min / median / max over `1 + n_alloc` allocations beside the first one's number.
    first_ms: the timed steps' per-launch time on the first allocation."""
    import statistics
    outs, ms = [], [first_ms]
    rows = pipeline.output_bytes(tb)
    for _ in range(n_alloc):
        outs.append(torch.empty(rows + 64, dtype=torch.uint8, device=tb["ops"].device))
    for out in outs:
        job = pipeline.Paf2MafStatJob(eng, tb, out=out)
        job.bind_stream()
        for _ in range(max(2, warm)):    # first touch of this buffer
            job.step()
        torch.cuda.synchronize()
        eng.expand_timing()
        for _ in range(steps):
            job.step()
        torch.cuda.synchronize()
        ms_sum, n_timed = eng.expand_timing()
        assert bool((job.diag == -1).all())
        ms.append(ms_sum / max(1, n_timed))
        ab = job.algorithmic_bytes()["expand"]
        del job
    del outs
    torch.cuda.empty_cache()
    fr = lambda t: ab / (t * 1e-3) / 1e9 / HBM_PEAK_GBS
    return {"allocations": len(ms), "k_ms_by_allocation": ms, "k_ms_min": min(ms), "k_ms_median": statistics.median(ms),
            "k_ms_max": max(ms), "frac_min": fr(max(ms)), "frac_median": fr(statistics.median(ms)), "frac_max": fr(min(ms)),
            "launches_per_allocation": max(2, warm) + steps,
            "note": "allocation 0 = the timed steps' buffer (the headline); the others are fresh buffers held side by side, with as "
                    "many warm-up and timed launches each as the headline: rocprofv3's average over one run of this command weighs "
                    "the five allocations equally"}


def e2e_leg(tb, synth, torch, check=2):
    """file to file on the same batch ("input GB/s is reported on the text size as well"): the records as a PAF
    file and the pools as two FASTA files under /tmp, then the `wgatools` command line (the C++ host layer over the C-ABI)
    runs `stat` and `paf2maf` on them — text parsing, PCIe both ways, kernels and the MAF written out — under its phase
    timer (WGA_TIMING=1).  Wall time of the whole process, HIP start-up included.  The first blocks of the MAF are compared
    with the oracle's rows."""
    import shutil, subprocess, tempfile
    from wgatools_amd import build
    cli = build.build_cli()
    rows_bytes = 2 * int((tb["mx"] + tb["i"] + tb["d"]).sum().item())
    tmp = tempfile.mkdtemp(prefix="wga_e2e_", dir="/tmp")
    try:
        free = shutil.disk_usage(tmp).free
        if free < rows_bytes * 1.6 + 4e9:
            return {"skipped": "%.0f GB free under /tmp, the MAF alone is %.0f GB" % (free / 1e9, rows_bytes / 1e9)}
        t_fa, q_fa, paf = (os.path.join(tmp, f) for f in ("t.fa", "q.fa", "in.paf"))
        for path, name, pool in ((t_fa, b"tchr", tb["t_pool"]), (q_fa, b"qchr", tb["q_pool"])):
            seq = pool.cpu().numpy().tobytes()
            with open(path, "wb") as f:
                f.write(b">" + name + b"\n")
                for i in range(0, len(seq), 1 << 20):
                    f.write(seq[i:i + (1 << 20)] + b"\n")
        text = synth.paf_text_torch(tb).cpu().numpy()
        text.tofile(paf)
        paf_bytes = int(text.size)
        del text
        out = {"records": tb["n"], "ops": tb["n_ops"], "paf_text_bytes": paf_bytes,
               "fasta_bytes": int(tb["t_pool"].numel() + tb["q_pool"].numel()),
               "note": "wall time of the wgatools process, file to file under /tmp (page cache), HIP start-up included"}
        # a profiler wrapped around bench.py (rocprofv3 --kernel-trace --stats -- python bench.py ...) stays with this process:
        # the command line's launches of the same kernels would otherwise be counted into the timed steps' statistics
        env = {k: v for k, v in os.environ.items()
               if not (k.startswith("ROCPROF") or k.startswith("ROCP_") or k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE"))}
        env["WGA_TIMING"] = "1"
        for name, argv, outp in (("stat", ["stat", "-f", "paf", paf], os.path.join(tmp, "out.tsv")),
                                 ("paf2maf", ["paf2maf", paf, "-g", t_fa, "-q", q_fa], os.path.join(tmp, "out.maf")),
                                 ("paf2maf_gz", ["paf2maf", paf, "-g", t_fa, "-q", q_fa], os.path.join(tmp, "out.maf.gz"))):
            t0 = time.perf_counter()
            r = subprocess.run([cli] + argv + ["-o", outp, "-r"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                               timeout=900)
            dt = time.perf_counter() - t0
            if r.returncode:
                out[name] = {"error": r.stderr.decode()[-300:]}
                continue
            ob = os.path.getsize(outp)
            phases = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("[timing]")]
            out[name] = {"wall_s": dt, "ops_per_s": tb["n_ops"] / dt, "input_text_GBps": paf_bytes / dt / 1e9,
                         "output_bytes": ob, "output_GBps": ob / dt / 1e9, "phases": phases[-1] if phases else None}
        if check and "wall_s" in out.get("paf2maf", {}):
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import parity_cases as pc
            with open(os.path.join(tmp, "out.maf"), "rb") as f:
                head = f.read(64 << 20)
            blocks = head.split(b"\n\n")
            for i in range(check):
                lines = blocks[i].split(b"\n")
                srows = [ln for ln in lines if ln.startswith(b"s\t")]
                et, eq = pc.oracle_rows(synth.torch_batch_record_to_numpy(tb, i), 0)
                assert srows[0].split(b"\t")[-1] == et and srows[1].split(b"\t")[-1] == eq, "MAF block %d differs from the oracle" % i
            out["paf2maf"]["parity_spot_check"] = "first %d MAF blocks bit-identical to oracle rows" % check
        if "wall_s" in out.get("paf2maf_gz", {}) and "wall_s" in out.get("paf2maf", {}):
            # `-o out.maf.gz` (utils.rs:201-209): BGZF members deflated on the device (K18).  Every member's ISIZE is added up
            # against the plain file's size; the first 64 members are inflated by zlib and compared with the plain file's head.
            import mmap, struct, zlib
            g = out["paf2maf_gz"]
            with open(os.path.join(tmp, "out.maf.gz"), "rb") as f, open(os.path.join(tmp, "out.maf"), "rb") as fp:
                mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
                p, total, members, head = 0, 0, 0, []
                while p < len(mm):
                    bsize = struct.unpack_from("<H", mm, p + 16)[0] + 1
                    total += struct.unpack_from("<I", mm, p + bsize - 4)[0]
                    if members < 64:
                        head.append(zlib.decompress(mm[p + 18:p + bsize - 8], -15))
                    members += 1
                    p += bsize
                head = b"".join(head)
                ok = p == len(mm) and total == out["paf2maf"]["output_bytes"] and fp.read(len(head)) == head
                mm.close()
            g["members"] = members
            g["ratio"] = g["output_bytes"] / out["paf2maf"]["output_bytes"]
            g["check"] = ("ISIZE of all members adds up to the plain MAF's size; the first 64 members inflate (zlib) to its head" if ok
                          else "MISMATCH against the plain MAF")
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def north_star(args):
    """The headline shape of BASELINE.json (`10 M records x mean 50 kop`; 2 TB of packed ops cannot be one resident batch):
    a stream of on-device-generated resident batches (`--ns-batch-records` x mean 50 kop each, <= 64 GB with its rows), every
    batch through K1 + layout + K2 once, the per-batch kernel times summed (generation is not timed: the records exist in HBM
    when a batch's timed region starts, as in the default line).  `--ns-records` sets how far the stream goes (default 400 000
    records = 2e10 ops, about a minute; the full 10 M take ~25 x that)."""
    import torch
    from wgatools_amd import engine, pipeline, synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    eng = engine.Engine(0)
    eng.set_param("expand_timing", 1)
    per = args.ns_batch_records
    nb = (args.ns_records + per - 1) // per
    tot_ops = tot_cols = tot_bytes = 0
    ms_step = ms_k2 = 0.0
    arena = None
    for b in range(nb):
        tb = synth.make_paf_batch_torch(0x5747415F + 1000 + b, per, 50_000, args.pool_mb * 1_000_000, dev)
        need = int((tb["t_src_len"] + tb["i"]).sum() + (tb["q_src_len"] + tb["d"]).sum()) + 64
        fresh = arena is None or arena.numel() < need
        if fresh:                       # one output arena for the whole stream, as a long-lived caller keeps: the first
            arena = None                # buffer the allocator returns
            torch.cuda.empty_cache()
            arena = torch.empty(int(need * 1.08), dtype=torch.uint8, device=dev)
        job = pipeline.Paf2MafStatJob(eng, tb, out=arena)
        job.bind_stream()
        if fresh:
            for _ in range(2):          # first touch of the arena
                job.step()
            torch.cuda.synchronize()
            eng.expand_timing()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        job.step()
        e1.record()
        torch.cuda.synchronize()
        assert bool((job.diag == -1).all()), "kernel reported per-record errors on clean synthetic input"
        k2, n_t = eng.expand_timing()
        ms_step += e0.elapsed_time(e1)
        ms_k2 += k2
        tot_ops += job.n_ops
        tot_cols += int((tb["mx"] + tb["i"] + tb["d"]).sum())
        tot_bytes += job.algorithmic_bytes()["expand"]
        del job, tb
        torch.cuda.empty_cache()
    ach = tot_bytes / (ms_k2 * 1e-3) / 1e9
    print(json.dumps({
        "metric": "paf_cigar_ops_per_s (paf2maf+stat)", "value": tot_ops / (ms_step * 1e-3), "unit": "ops/s", "n_gpus": 1,
        "steps": nb, "warmup": 5, "ms_per_step": ms_step / nb, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "north-star shape: %d records x mean 50 kop streamed as %d on-device-generated resident batches of %d "
                               "records (2 x %d Mb pools); the full headline is 10 000 000 records" % (nb * per, nb, per, args.pool_mb),
                   "records": nb * per, "ops": tot_ops, "columns": tot_cols},
        "roofline": {"kernel": {0: "k_paf2maf_expand", 3: "k_paf2maf_expand_s"}.get(eng.get_param("expand_variant_used")),
                     "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": tot_bytes},
        "output_placement": "first allocation",
        "metric_scope": "kernel-only, summed over the batches, rows written into one output arena (the first buffer the allocator "
                        "returns); generation between batches is not timed",
    }), flush=True)
    eng.close()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, the environment the driver's
    torch.distributed.run line would give them) and pass rank 0's line through.  Fewer than N devices: exit code 2."""
    import subprocess
    try:
        import torch
        have = torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        have = 0
    if have < n:
        sys.stderr.write("bench.py: --gpus %d but %d device(s) visible\n" % (n, have))
        raise SystemExit(2)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    raise SystemExit(rc)


def strong_plan(args, world):
    """The strong-scaling job: ONE global batch of --records over the targets of a 64-way all-to-all (PanSN names, human-like
    contig sizes, records in proportion to the contig's size; or --targets equal-sized names), and who owns which target under
    both assignment rules.  Pure host work (numpy): `--plan-only` prints it without a GPU."""
    import numpy as np
    from wgatools_amd import multigpu, synth
    gseed = 0x5747415F + 2
    n_ops_all = synth.record_lengths(gseed, args.records, args.mean_ops)
    rng = np.random.default_rng(gseed + 1)
    if args.targets > 0:
        names = ["g%02d#1#chr1" % k for k in range(args.targets)]
        p = 1.0 / np.arange(1, args.targets + 1) ** args.zipf
        what = "%d equal-sized targets (zipf %.2f)" % (args.targets, args.zipf)
    else:
        names, mb = multigpu.human_like_targets(args.genomes)
        p = mb / np.arange(1, len(names) + 1) ** args.zipf
        what = "%d assemblies x chr1 .. chrY (PanSN names, records in proportion to the contig's size%s)" % (
            args.genomes, ", zipf %.2f" % args.zipf if args.zipf else "")
    tid = rng.choice(len(names), size=args.records, p=p / p.sum())
    rec_names = [names[k] for k in tid]
    own = {"hash": multigpu.owners(rec_names, world), "lpt": multigpu.owners_lpt(rec_names, n_ops_all, world)}
    rep = {"targets": what, "ranks": world, "assign": args.assign}
    for k, o in own.items():
        per = np.bincount(o, weights=n_ops_all.astype(np.float64), minlength=world)
        rep["ops_per_rank_" + k] = [float(x) for x in per]
        rep["imbalance_max_over_mean_" + k] = float(per.max() / per.mean()) if per.mean() > 0 else 1.0
    return {"owner": own[args.assign], "n_ops_all": n_ops_all, "report": rep}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps (the first launch touches the output buffer for the first time)")
    ap.add_argument("--records", type=int, default=100_000)
    ap.add_argument("--mean-ops", type=int, default=5000)
    ap.add_argument("--pool-mb", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", type=int, default=8, help="records spot-checked against the oracle")
    ap.add_argument("--param", action="append", default=[], help="engine test knob name=value")
    ap.add_argument("--neg-frac", type=float, default=0.5)
    ap.add_argument("--m-only", action="store_true", help="variant of configs[1] with = / X merged into M ops")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group (RCCL) and run the collectives of the N > 1 "
                    "path at world size 1 too")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file-to-file leg (the wgatools command line on the same batch)")
    ap.add_argument("--no-extras", action="store_true", help="skip the genome-sized-pool and 50-kop-record K2 measurements")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): ONE global batch of --records whatever N — the file-shaped job: every rank takes the "
                         "records fnv1a64(target_name) %% N gives it, ordered output through one all-reduce of the record sizes; "
                         "weak: --records per GPU, independently generated shards")
    ap.add_argument("--targets", type=int, default=0, help="strong scaling: number of equal-sized target names gNN#1#chr1 (0, the "
                                                           "default: the 64 x 24 contigs of 64 human-like assemblies, chr1 .. chrY at "
                                                           "their sizes, records in proportion to the contig's size)")
    ap.add_argument("--genomes", type=int, default=64, help="strong scaling: assemblies of the default target set")
    ap.add_argument("--assign", choices=["hash", "lpt"], default="hash",
                    help="strong scaling: which rank owns a target — hash: fnv1a64(target_name) %% N, the product's rule (no table, no "
                         "pass over the input); lpt: targets by their total ops, heaviest first, each to the lightest rank so far.  "
                         "The line reports the imbalance of both")
    ap.add_argument("--plan-only", action="store_true", help="print the strong-scaling partition of --gpus ranks (ops per rank and "
                                                             "imbalance under both assignments) and exit: needs no GPU")
    ap.add_argument("--zipf", type=float, default=0.0, help="strong scaling: skew of the records over the targets "
                                                            "(P(target k) ~ 1 / (k + 1)^zipf; 0 = uniform, the default)")
    ap.add_argument("--no-placement-probe", action="store_true", help="(accepted for older command lines: the output buffer is the first allocation)")
    ap.add_argument("--no-spread", action="store_true", help="skip the row kernel's times on four further fresh output allocations "
                                                             "(roofline.allocation_spread)")
    ap.add_argument("--north-star", action="store_true", help="the 10 M x 50 kop headline shape as a stream of resident batches (N = 1)")
    ap.add_argument("--ns-records", type=int, default=400_000)
    ap.add_argument("--ns-batch-records", type=int, default=40_000)
    args = ap.parse_args()
    if args.north_star:
        return north_star(args)
    if args.plan_only:
        return print(json.dumps(strong_plan(args, max(1, args.gpus))["report"]), flush=True)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)     # a bare `python bench.py --gpus N`: this process starts the N ranks itself

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE %d: the line reports n_gpus = %d\n" % (args.gpus, world, world))
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d wants device %d, %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # --force-dist: the process group and every collective of the N > 1 path also at world size 1 (so that the RCCL branches
    # have run on a one-GPU box; the driver's N > 1 runs take them anyway)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from wgatools_amd import engine, multigpu, pipeline, synth
    eng = engine.Engine(local_rank)
    for kv in args.param:
        k, v = kv.split("=")
        eng.set_param(k, int(v))
    seed = 0x5747415F + 2 + rank
    strong = args.scaling == "strong" and dist_on   # at N = 1 the global batch IS the rank's batch: nothing to shard or order
    if strong:
        # ONE global batch whatever N: record lengths and target names from the global seed; a rank generates the records
        # fnv1a64(target_name) % N gives it (the product's sharding rule), so the skew over targets becomes load imbalance
        plan = strong_plan(args, world)
        n_ops_all, owner = plan["n_ops_all"], plan["owner"]
        mine = multigpu.my_records(owner, rank)
        tb = synth.make_paf_batch_torch(seed, len(mine), args.mean_ops, args.pool_mb * 1_000_000, dev,
                                        neg_frac=args.neg_frac, use_m=args.m_only, n_ops=n_ops_all[mine])
        mine_idx = torch.as_tensor(mine, dtype=torch.int64, device=dev)
        sizes_global = torch.zeros(args.records, dtype=torch.int64, device=dev)
    else:
        tb = synth.make_paf_batch_torch(seed, args.records, args.mean_ops, args.pool_mb * 1_000_000, dev,
                                        neg_frac=args.neg_frac, use_m=args.m_only)
    # The timed steps run on the FIRST output buffer the allocator returns: what any caller gets.
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
        if strong:
            # ordered output without a gather: every record's byte count is known now, before a row byte exists; one
            # all-reduce of the global size vector gives each rank the final (input-order) offsets of its records
            sizes_global.zero_()
            sizes_global[mine_idx] = job.rec_off[1:] - job.rec_off[:-1]
            if dist_on:
                dist.all_reduce(sizes_global)
            global_off = torch.cumsum(sizes_global, 0) - sizes_global   # noqa: F841  (what the writer would use)
        if evs:
            evs[2].record()
        job.expand()
        if evs:
            evs[3].record()
        eng.counts_total(job.n, job.counts, totals)          # global stat totals (88 bytes)
        if dist_on:
            dist.all_reduce(totals)                          # RCCL over xGMI: 88 bytes

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events inside the library bracket the dominant kernel alone (on the launch stream)
    eng.set_param("expand_timing", 1)
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    nops = torch.tensor([float(job.n_ops)], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(nops)
    elapsed = float(el.item())
    total_ops = float(nops.item())

    ranks_seen = None
    if dist_on:   # every rank adds one: what RCCL itself saw
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    # the run only counts if every record came out clean
    if not args.param:
        assert bool((job.diag == -1).all()), "kernel reported per-record errors on clean synthetic input"

    per_rank_ops, imb = multigpu.imbalance(job.n_ops, dist if dist_on else None, dev)
    # N > 1: every rank checks a few of the rows ITS timed steps wrote against the oracle (the checker leg of the line: the only
    # use bench.py makes of oracle/ besides the N = 1 cpu baseline), and the ranks agree on the verdict — an N-GPU line is
    # a real partition AND byte-verified
    ranks_checked = None
    if dist_on and args.check and not args.param:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import parity_cases as pc
        ok = 1
        for i in torch.linspace(0, tb["n"] - 1, min(args.check, tb["n"])).long().tolist():
            r = synth.torch_batch_record_to_numpy(tb, i)
            et, eq = pc.oracle_rows(r, 0)
            gt, gq = job.record_rows(i)
            ok &= int(gt == et and gq == eq)
        okt = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        assert int(okt.item()) == 1, "a rank's rows differ from the oracle"
        ranks_checked = "%d records on each of %d ranks bit-identical to oracle rows" % (min(args.check, tb["n"]), world)
    if rank == 0:
        k_stat = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
        k_layout = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps
        k_expand_call = sum(e[2].elapsed_time(e[3]) for e in events) / args.steps   # pre-pass + kernel
        ms_sum, n_timed = eng.expand_timing()
        k_expand = ms_sum / n_timed if n_timed else k_expand_call
        variant_used = eng.get_param("expand_variant_used")
        kernel_name = {0: "k_paf2maf_expand", 3: "k_paf2maf_expand_s"}.get(variant_used, "k_paf2maf_expand")
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: paf2maf+stat, %d records%s x mean %d ops "
                            "(lognormal s=0.5; =/X/I/D mix), 2 x %d Mb pools, strand 50/50, "
                            "HBM-resident" % (args.records, "" if args.scaling == "strong" else "/GPU", args.mean_ops, args.pool_mb),
                "records_per_gpu": args.records, "ops_per_gpu": job.n_ops,
                "columns_per_gpu": int((tb["mx"] + tb["i"] + tb["d"]).sum()),
                "output_bytes_per_gpu": job.out_bytes,
                "sharding": ("one global batch of %d records, a target's records on one rank (--assign %s); "
                             "per step one all-reduce of the per-record output sizes (ordered output) and one of the stat "
                             "totals; row bytes never cross GPUs" % (args.records, args.assign)) if strong
                            else "records, no data-path collective",
                "ops_per_rank": per_rank_ops, "imbalance_max_over_mean": imb,
                "partition": plan["report"] if strong else None,
                "ranks_seen_by_rccl": ranks_seen,
            },
            "input_GBps": in_bytes * world * args.steps / elapsed / 1e9,
            "kernel_ms": {"k_cigar_stat": k_stat, "layout_scan": k_layout, kernel_name: k_expand,
                          "expand_prepass (k_rec_desc + k_tile_base + marks)": k_expand_call - k_expand},
            "row_kernel": {"expand_variant_used": variant_used, "name": kernel_name,
                           "tiles_left_to_v1": eng.get_param("expand_stream_left_to_v1") if variant_used == 3 else None},
            "output_placement": "first allocation",
            "roofline": {
                "kernel": kernel_name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(args, job),
                "algorithmic_bytes_per_launch": ab["expand"],
                "output_buffer": "first allocation (no placement policy)",
                "k_cigar_stat_GBps": ab["stat"] / (k_stat * 1e-3) / 1e9,
                # paf2maf+stat priced both ways.  The step runs K1 and K2 as two kernels, so the packed ops
                # are read twice (unfused); a fused walk would read them once.
                "paf2maf_plus_stat": {
                    "fused": False,
                    "bytes_unfused": ab["stat"] + ab["expand"], "bytes_fused_minimum": ab["expand"] + 88 * job.n,
                    "achieved_GBps_unfused_bytes": (ab["stat"] + ab["expand"]) / ((k_stat + k_expand) * 1e-3) / 1e9,
                    "frac_unfused_bytes": (ab["stat"] + ab["expand"]) / ((k_stat + k_expand) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "frac_fused_minimum_bytes": (ab["expand"] + 88 * job.n) / ((k_stat + k_expand) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "K1 + K2 kernel time; ops (4 B each) read by both kernels",
                },
            },
        }
        if ranks_checked:
            result["parity_spot_check"] = ranks_checked
        result["metric_scope"] = ("kernel-only: K1 + layout + K2 (+ totals) on packed ops and sequence pools resident in HBM; no "
                                  "CIGAR tokenising, PAF / FASTA parsing, PCIe or file I/O — the file-to-file command line on the same "
                                  "batch is the `e2e` object of this line")
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
        if world == 1 and not args.no_spread and not args.param:
            try:   # additional information: never at the price of the headline line
                result["roofline"]["allocation_spread"] = allocation_spread(eng, pipeline, torch, tb, k_expand, steps=args.steps,
                                                                            warm=args.warmup)
            except Exception as e:  # noqa: BLE001
                result["roofline"]["allocation_spread"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_extras and not args.param and args.records == 100_000 and args.mean_ops == 5000:
            # the same kernel where the default shape flatters it: pools beyond the Infinity Cache, and the
            # north-star record length
            try:   # additional information: never at the price of the headline line
                del job
                torch.cuda.empty_cache()
                ms_g, frac_g = extra_shape(eng, synth, pipeline, torch, dev, seed, args.records, args.mean_ops, 1000)
                ms_l, frac_l = extra_shape(eng, synth, pipeline, torch, dev, seed + 100, 10_000, 50_000, args.pool_mb)
                ms_lg, frac_lg = extra_shape(eng, synth, pipeline, torch, dev, seed + 100, 10_000, 50_000, 1000)
                ms_s, frac_s = extra_shape(eng, synth, pipeline, torch, dev, seed + 200, 1_000_000, 500, args.pool_mb)
                result["roofline"]["frac_genome_pools"] = frac_g
                result["roofline"]["frac_50kop_records"] = frac_l
                result["roofline"]["frac_50kop_records_genome_pools"] = frac_lg
                result["roofline"]["frac_500op_records"] = frac_s     # short records: many row pieces per tile (0.34-0.36 so far)
                result["roofline"]["extra_shapes_ms"] = {"2x1GB_pools": ms_g, "10000x50kop": ms_l, "10000x50kop_2x1GB_pools": ms_lg,
                                                         "1000000x500op": ms_s}
            except Exception as e:  # noqa: BLE001
                result["roofline"]["extra_shapes_error"] = "%s: %s" % (type(e).__name__, e)
        if world == 1 and not args.no_e2e and not args.param:
            try:   # additional information: never at the price of the headline line
                try:
                    del job
                except NameError:
                    pass
                torch.cuda.empty_cache()
                result["e2e"] = e2e_leg(tb, synth, torch, check=2 if args.check and not args.no_cpu_baseline else 0)
            except Exception as e:  # noqa: BLE001
                result["e2e"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(result), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
