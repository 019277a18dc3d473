"""A/B of the paf2maf row kernel (K2) in ONE process on the same buffers: the run-to-run spread of K2 is the physical
placement of its output buffer (profiles/r02_k2_experiments.md, section 7), so builds are only comparable on one allocation.

  locally:  python scripts/gpu_k2_same_buffers.py build NAME=FLAGS ...     (build_variants/libwgahip_NAME.so; "base=" = no flags)
  GPU box:  python scripts/gpu_k2_same_buffers.py run NAME[:param=val,...] ...  [--shape records,mean_ops,pool_mb]...

Each candidate = a library build plus context parameters (e.g. expand_variant=2).  Every round times every candidate
(6 launches each, library HIP events), rounds alternate the order; the table gives mean / min per candidate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "build":
    from wgatools_amd import build
    for a in sys.argv[2:]:
        name, flags = a.split("=", 1)
        print(build.build_hip_variant(name, flags.split()))
    sys.exit(0)

import torch
from wgatools_amd import build, engine, synth, pipeline, _lib

cands, shapes = [], []
args = sys.argv[2:]
k = 0
while k < len(args):
    if args[k] == "--shape":
        shapes.append(tuple(int(x) for x in args[k + 1].split(",")))
        k += 2
    else:
        name, _, ps = args[k].partition(":")
        cands.append((args[k], name, dict((p.split("=")[0], int(p.split("=")[1])) for p in ps.split(",") if p)))
        k += 1
shapes = shapes or [(100_000, 5000, 50)]
dev = torch.device("cuda", 0)
libs = {}
for _, name, _p in cands:
    if name not in libs:
        path = build.HIP_LIB if name == "tree" else os.path.join(build.ROOT, "build_variants", "libwgahip_%s.so" % name)
        libs[name] = _lib.load(path, require_all=(name == "tree"))   # builds of older commits lack the newer entry points
for rec, mean, pool in shapes:
    tb = synth.make_paf_batch_torch(0x5747415F + 2, rec, mean, pool * 1_000_000, dev)
    jobs = []
    out = None
    for label, name, params in cands:
        eng = engine.Engine(0, libs[name])
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        for pk, pv in params.items():
            eng.set_param(pk, pv)
        eng.set_param("expand_timing", 1)
        job = pipeline.Paf2MafStatJob(eng, tb, out=out)
        out = job.out                      # every candidate writes the same bytes of HBM
        job.bind_stream()
        job.stat(); job.layout(); job.expand(); torch.cuda.synchronize()
        jobs.append((label, eng, job, []))
    for rnd in range(6):
        order = jobs if rnd % 2 == 0 else jobs[::-1]
        for label, eng, job, acc in order:
            eng.expand_timing()
            for _ in range(6):
                job.expand()
            torch.cuda.synchronize()
            ms, n = eng.expand_timing()
            acc.append(ms / n)
    print("shape %d x %d op, 2 x %d MB pools (one output buffer @ 0x%x):" % (rec, mean, pool, out.data_ptr()))
    base = sum(jobs[0][3]) / len(jobs[0][3])
    for label, eng, job, acc in jobs:
        m = sum(acc) / len(acc)
        print("  %-44s K2 mean %.3f ms  min %.3f  (%+.1f %% vs first)" % (label, m, min(acc), 100 * (m / base - 1)))
    for _l, eng, _j, _a in jobs:
        eng.close()
    del jobs, tb, out
    torch.cuda.empty_cache()
