"""K5 (pafcov) on the emulator over seeds the suite does not use — the checks of tests/parity_cases.py (random shapes through both
protocols, with rare ops far longer than a window, and every fourth seed a few of 2^27 .. 2^28 - 1 bases: WIDE pieces):
python scripts/emu_campaign_pafcov.py <first seed> <seconds>.  CPU only; results: profiles/r06_emu_campaign.txt."""
import sys, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import parity_cases as pc
from wgatools_amd import build, engine, _lib
eng = engine.Engine(0, _lib.load(build.EMU_LIB))
t0 = time.time()
fails = 0
seed0 = int(sys.argv[1]); budget = float(sys.argv[2])
k = 0
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    try:
        pc.check_pafcov_random(eng, s, 2)
        if s % 4 == 0:      # wide pieces: N / D ops of 2^27 .. 2^28 - 1 bases inside short records, several per tile
            rng = np.random.default_rng(s)
            n = int(rng.integers(3, 60))
            recs = []
            for _ in range(n):
                m = int(rng.integers(1, 40))
                c = rng.choice(np.array([7, 7, 0, 8, 1, 2, 3], dtype=np.uint32), m)
                ln = rng.integers(0, 50, m).astype(np.uint32)
                if rng.random() < 0.5:
                    j = int(rng.integers(0, m))
                    c[j] = rng.choice(np.array([3, 2, 7], dtype=np.uint32))
                    ln[j] = int(rng.integers(1 << 27, 1 << 28))
                recs.append((ln << 4) | c)
            b = dict(ops=np.concatenate(recs).astype(np.uint32), op_off=np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64),
                     strand_neg=np.zeros(n, dtype=np.uint8))
            nt = int(rng.integers(1, 4))
            tlen = [int(rng.integers(1, 200000)) for _ in range(nt)]
            tid = rng.integers(0, nt, n)
            ts = [int(rng.integers(0, tlen[t] + 1)) for t in tid]
            pc.check_pafcov(eng, b, list(tid), ts, tlen, align=int(rng.choice([1, 4])), split=bool(rng.integers(0, 2)))
    except Exception:
        fails += 1
        print("FAIL pafcov", s); traceback.print_exc(); sys.stdout.flush()
    if k % 20 == 0:
        print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
print("END", k, "seeds", fails, "fails")
