"""K11 (runs -> ops / cg:Z: text through the block table of round 6) and K9 (BED lines staged in LDS) on the emulator over seeds the
suite does not use: python scripts/emu_campaign_bridges.py <first seed> <seconds>.  CPU only; results: profiles/r06_emu_campaign.txt."""
import sys, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import parity_cases as pc
from wgatools_amd import build, engine, _lib
eng = engine.Engine(0, _lib.load(build.EMU_LIB))
t0 = time.time(); fails = 0
seed0 = int(sys.argv[1]); budget = float(sys.argv[2]); k = 0
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    rng = np.random.default_rng(s)
    try:
        pc.check_bridge_blocks(eng, seed=s, n=int(rng.integers(1, 60)))
        name = bytes(rng.choice(np.frombuffer(b"abcXYZ#_.0123", dtype=np.uint8), int(rng.choice([0, 1, 4, 10, 33, 70]))))
        cnt = int(rng.choice([1, 2, 255, 256, 257, 511, 512, 513, 1024, 1500, 2100]))
        p0 = int(rng.choice([0, 7, 95, 999_990, 99_999_000, 999_999_500, 4_294_966_800, 9_999_999_500]))
        pc.check_pafcov_format(eng, name, rng.integers(0, int(rng.choice([2, 11, 1000, 2_000_000_000])), cnt), p0)
    except Exception:
        fails += 1
        print("FAIL", s); traceback.print_exc(); sys.stdout.flush()
    if k % 10 == 0:
        print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
print("END", k, "seeds", fails, "fails")
