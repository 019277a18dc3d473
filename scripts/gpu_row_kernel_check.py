"""A row kernel (expand_variant 3 = the streaming kernel; argv[1]) on the GPU: the oracle
battery, and the whole output of BASELINE configs[1] / 50-kop / 500-op batches byte for byte against v1 (expand_variant 0)
on the device.  Further arguments: context parameters name=value (e.g. expand_job_tiles=8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import parity_cases as pc
from wgatools_amd import build, engine, _lib, synth, pipeline

eng = engine.Engine(0, _lib.load(build.HIP_LIB))
V = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for kv in sys.argv[2:]:
    eng.set_param(kv.split("=")[0], int(kv.split("=")[1]))
t0 = time.time()
for seed, n, mean, pool, use_m in [(1, 12, 700, 50000, False), (2, 40, 60, 20000, False), (3, 300, 3, 5000, True), (4, 3, 5000, 200000, True)]:
    b = synth.make_paf_batch(seed, n, mean, pool, use_m=use_m)
    rng = np.random.default_rng(seed)
    pc.check_paf2maf(eng, b, variant=V)
    pc.check_paf2maf(eng, b, pre=(rng.integers(0, 40, n), rng.integers(0, 40, n), rng.integers(0, 5, n)), variant=V)
e = pc.edge_case_batch(eng)
ne = len(e["strand_neg"])
rng = np.random.default_rng(5)
pc.check_paf2maf(eng, e, variant=V)
pc.check_paf2maf(eng, e, pre=(rng.integers(0, 33, ne), rng.integers(0, 33, ne), rng.integers(0, 3, ne)), variant=V)
pc.check_paf2maf(eng, pc.wide_tile_batch(eng), variant=V)
pc.check_paf2maf(eng, e, force_slow=1, variant=V)
for seed in range(20, 24):
    pc.check_paf2maf(eng, synth.make_paf_batch(seed, 700, 2, 8000), variant=V)
for seed in range(30, 46):
    n = int(np.random.default_rng(seed).integers(1, 300))
    mean = int(np.random.default_rng(seed + 1).integers(1, 3000))
    b = synth.make_paf_batch(seed, n, mean, 300000, use_m=bool(seed & 1))
    rng = np.random.default_rng(seed)
    pc.check_paf2maf(eng, b, pre=(rng.integers(0, 130, n), rng.integers(0, 130, n), rng.integers(0, 5, n)), variant=V)
pc.check_paf2maf(eng, synth.make_paf_batch(6, 500, 400, 2_000_000), variant=V)
pc.check_paf2maf(eng, synth.make_paf_batch(12, 1, 300_000, 1_500_000, sigma=0.01), variant=V)
print("oracle battery ok (%.1f s)" % (time.time() - t0), flush=True)

dev = torch.device("cuda", 0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for rec, mean, pool in [(100_000, 5000, 50), (10_000, 50_000, 50), (1_000_000, 500, 50), (3_000_000, 30, 20)]:
    tb = synth.make_paf_batch_torch(0x5747415F + 2, rec, mean, pool * 1_000_000, dev)
    outs = []
    for v in (0, V):
        eng.set_param("expand_variant", v)
        job = pipeline.Paf2MafStatJob(eng, tb, with_text=True)
        job.out.fill_(0x23)
        job.bind_stream()
        job.step()
        torch.cuda.synchronize()
        assert bool((job.diag == -1).all())
        outs.append(job.out)
        del job
    same = bool(torch.equal(outs[0], outs[1]))
    print("shape %d x %d, 2 x %d MB: variant %d == variant 0 over %d bytes: %s" % (rec, mean, pool, V, outs[0].numel(), same), flush=True)
    if not same:
        d = (outs[0] != outs[1]).nonzero()[:10].flatten().tolist()
        print("  first differences at", d)
        sys.exit(1)
    del outs, tb
    torch.cuda.empty_cache()
eng.set_param("expand_variant", 0)
print("row kernel %d check ok" % V)
