"""The emulator checks of tests/parity_cases.py (stat, the four row-kernel choices, pafcov, the paf2maf -> maf2paf round trip) on seeds
the suite does not use: python scripts/emu_campaign_rows_cov.py <first seed> <seconds>.  CPU only; results: profiles/r05_emu_campaign.txt."""
import sys, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import parity_cases as pc
from wgatools_amd import build, engine, _lib, synth
eng = engine.Engine(0, _lib.load(build.EMU_LIB))
t0 = time.time()
fails = 0
def attempt(name, fn):
    global fails
    try:
        fn()
    except Exception:
        fails += 1
        print("FAIL", name); traceback.print_exc(); sys.stdout.flush()
seed0 = int(sys.argv[1]); budget = float(sys.argv[2])
k = 0
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    rng = np.random.default_rng(s)
    n = int(rng.integers(1, 60)); mean = int(rng.choice([3, 20, 90, 110, 400, 1500])); pool = int(rng.choice([3000, 60000, 400000]))
    use_m = bool(rng.integers(0, 2))
    b = synth.make_paf_batch(s, n, mean, pool, use_m=use_m)
    attempt("stat %d" % s, lambda: pc.check_stat(eng, b))
    for variant in (-1, 0, 3):
        pre = (rng.integers(0, 40, n), rng.integers(0, 40, n), rng.integers(0, 5, n)) if rng.integers(0, 2) else None
        attempt("paf2maf %d v%d" % (s, variant), lambda: pc.check_paf2maf(eng, b, pre=pre, variant=variant))
    attempt("pafcov %d" % s, lambda: pc.check_pafcov_random(eng, s, 2))
    attempt("roundtrip %d" % s, lambda: pc.check_paf2maf_maf2paf_roundtrip(eng, s, min(n, 20), min(mean, 400)))
    print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
print("END", k, "seeds", fails, "fails")
