"""K5 (pafcov accumulate) against the size of ONE resident batch: does the time per op stay flat when the op stream grows
from 10 GB to 100 GB?  (round 3: one call over 104 GB of ops measured 4.8 s, ten calls over 10 GB each 0.37 s.)
usage: [K5_MODE=sep|fused|both] [K5_REPS=n] python scripts/gpu_k5_scaling.py [chunks ...]   (2 M records of ~1300 ops per chunk;
10 chunks = configs[3] at its stated size)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wgatools_amd import build, engine, _lib, synth

dev = torch.device("cuda", 0)
eng = engine.Engine(0, _lib.load(os.environ.get("WGA_LIB") or build.HIP_LIB))     # WGA_LIB: a build_variants library
eng.set_stream(torch.cuda.current_stream().cuda_stream)
nt, tlen, per = 64, 100_000_000, 2_000_000
cov_len = torch.full((nt,), tlen, dtype=torch.int64, device=dev)
cov_off = torch.arange(nt, device=dev, dtype=torch.int64) * (tlen + 4)
total = int(nt * (tlen + 4))
cov = torch.zeros(total + 8, dtype=torch.int32, device=dev)
for chunks in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    n_all = chunks * per
    cap = int(n_all * 1300 * 1.06)
    ops = torch.empty(cap, dtype=torch.int32, device=dev)
    op_off = torch.zeros(n_all + 1, dtype=torch.int64, device=dev)
    strand = torch.zeros(n_all, dtype=torch.uint8, device=dev)
    t_start = torch.zeros(n_all, dtype=torch.int64, device=dev)
    target_id = torch.zeros(n_all, dtype=torch.int32, device=dev)
    n_ops = 0
    for k in range(chunks):
        tb = synth.make_paf_batch_torch(400 + k, per, 1300, tlen, dev)
        g = torch.Generator(device=dev)
        g.manual_seed(900 + k)
        ops[n_ops:n_ops + tb["n_ops"]] = tb["ops"]
        op_off[k * per + 1:(k + 1) * per + 1] = tb["op_off"][1:] + n_ops
        strand[k * per:(k + 1) * per] = tb["strand_neg"]
        t_start[k * per:(k + 1) * per] = tb["t_src_off"]
        target_id[k * per:(k + 1) * per] = torch.randint(0, nt, (per,), device=dev, generator=g, dtype=torch.int32)
        n_ops += tb["n_ops"]
        del tb
        torch.cuda.empty_cache()
    batch = engine.Batch(ops, op_off, strand, n_all, n_ops)
    mode = os.environ.get("K5_MODE", "both")        # sep: accumulate + finalize; fused: accumulate_final; both
    reps = int(os.environ.get("K5_REPS", "2"))
    ev = lambda: torch.cuda.Event(enable_timing=True)
    ref = None
    if os.environ.get("K5_COLD"):   # what the first call's 4.7 s are made of: the library's work buffers through hipMalloc (tile slots
        import time                 # nt x 8 x 32 B, scratch, pieces: ~12 GB at this size), timed here on allocations of those sizes
        import numpy as np
        nt_tiles = (n_ops + 1023) // 1024
        for name, nbytes in (("tile slots", nt_tiles * 8 * 32), ("scratch", nt_tiles * 84 + n_all * 16), ("pieces", int(nt_tiles * 2.8 * 1.25) * 32)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x = eng.empty(int(nbytes), np.uint8)
            eng.sync()
            t1 = time.perf_counter()
            keep = os.environ["K5_COLD"] == "2"       # 2: the three stay allocated side by side (fresh memory every time), freed at the end
            if keep:
                held = globals().setdefault("_held", [])
                held.append(x)
            else:
                x.free()
            eng.sync()
            print("hipMalloc of %.2f GB (%s): %.0f ms%s" % (nbytes / 1e9, name, (t1 - t0) * 1e3, "" if keep else ", free %.0f ms" % ((time.perf_counter() - t1) * 1e3)), flush=True)
        for x in globals().get("_held", []):
            x.free()
        globals()["_held"] = []
    for rep in range(reps):
        if mode in ("sep", "both"):
            cov.zero_()
            torch.cuda.synchronize()
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            eng.pafcov_accumulate(batch, target_id, t_start, cov_off, cov_len, cov, total)
            e1.record()
            eng.pafcov_finalize(nt, cov_off, cov_len, cov)
            e2.record()
            torch.cuda.synchronize()
            ms, msf = e0.elapsed_time(e1), e1.elapsed_time(e2)
            print("chunks %d (%.1f GB of ops, %d records) rep %d: accumulate %.1f ms = %.0f GB/s of op stream, finalize %.1f ms, "
                  "together %.1f ms" % (chunks, 4 * n_ops / 1e9, n_all, rep, ms, 4 * n_ops / ms / 1e6, msf, ms + msf), flush=True)
            if mode == "both" and rep == 0:
                ref = cov.clone()
        if mode in ("fused", "both"):
            cov.zero_()
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            eng.pafcov_accumulate_final(batch, target_id, t_start, cov_off, cov_len, nt, cov, total)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            alg = 4 * n_ops + 8 * nt * tlen
            print("chunks %d rep %d: accumulate_final %.1f ms = %.0f GB/s over 4n + 8 B per counter (%.1f GB)" % (
                chunks, rep, ms, alg / ms / 1e6, alg / 1e9), flush=True)
            if ref is not None:
                print("  fused == accumulate + finalize on every counter:", bool(torch.equal(ref, cov)), flush=True)
                ref = None
    del ops, op_off, strand, t_start, target_id, batch
    torch.cuda.empty_cache()
