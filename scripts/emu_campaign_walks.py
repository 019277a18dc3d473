"""MAF walks, call events, chain text, dotplot and pafpseudo on the emulator with seeds the suite does not use:
python scripts/emu_campaign_walks.py <first seed> <seconds>.  CPU only; results: profiles/r06_emu_campaign.txt."""
import sys, time, traceback
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import numpy as np
import parity_cases as pc
from wgatools_amd import build, engine, _lib, synth
eng = engine.Engine(0, _lib.load(build.EMU_LIB))
t0 = time.time(); fails = 0
def attempt(name, fn):
    global fails
    try:
        fn()
    except Exception:
        fails += 1
        print("FAIL", name); traceback.print_exc(); sys.stdout.flush()
seed0 = int(sys.argv[1]); budget = float(sys.argv[2]); k = 0
while time.time() - t0 < budget:
    s = seed0 + k; k += 1
    rng = np.random.default_rng(s)
    # MAF pairs of random shapes
    pairs, strands = [], []
    for _ in range(int(rng.integers(1, 40))):
        L = int(rng.choice([0, 1, 15, 16, 17, 63, 64, 65, 300, 1023, 1024, 1025, 1500, 2049, 5000]))
        L = max(0, L + int(rng.integers(-2, 3)))
        alpha = [b"ACGTacgt--N", b"ACGT-", b"AC-", b"ACGTN-acgtn"][int(rng.integers(0, 4))]
        t = pc.rand_seq(rng, L, alpha)
        q = bytearray(pc.rand_seq(rng, L + int(rng.integers(0, 3)) * int(rng.integers(0, 2)), alpha))
        if rng.random() < 0.7:
            for j in range(0, L, 4):
                if rng.random() < 0.85:
                    q[j:j + 4] = t[j:j + 4]
        pairs.append((t, bytes(q[:max(len(q), 0)]))); strands.append(int(rng.integers(0, 2)))
    # round 6: the stream walks' knobs at random — blocks per wave, where a block counts as long, the piece size (tiny pieces,
    # pieces of a few steps: the packed totals are folded every three steps in this build)
    grp, lng, pcs = int(rng.integers(0, 9)), int(rng.choice([32768, 100, 500, 2100])), int(rng.choice([16384, 17, 64, 1000, 2048, 7000]))
    eng.set_param("maf_group", grp); eng.set_param("maf_long_cols", lng); eng.set_param("maf_piece_cols", pcs)
    attempt("maf_pair %d (group %d long %d piece %d)" % (s, grp, lng, pcs), lambda: pc.check_maf_pair(eng, pairs, strands))
    attempt("maf_call %d (group %d long %d piece %d)" % (s, grp, lng, pcs), lambda: pc.check_maf_call_runs(eng, pairs))
    eng.set_param("maf_group", 0); eng.set_param("maf_long_cols", 32768); eng.set_param("maf_piece_cols", 16384)
    attempt("dotplot_maf %d" % s, lambda: pc.check_dotplot_maf(eng, pairs, strands, int(rng.integers(0, 50))))
    n = int(rng.integers(1, 40)); mean = int(rng.choice([3, 30, 200, 900]))
    b = synth.make_paf_batch(s, n, mean, 60000, use_m=bool(rng.integers(0, 2)))
    for svlen, snp in ((0, True), (int(rng.integers(1, 60)), bool(rng.integers(0, 2)))):
        attempt("paf_call %d" % s, lambda: pc.check_paf_call_events(eng, b["ops"], b["op_off"], svlen, snp))
    attempt("chain %d" % s, lambda: pc.check_cigar_chain(eng, b["ops"], b["op_off"]))
    attempt("dotplot %d" % s, lambda: pc.check_dotplot(eng, b["ops"], b["op_off"], b["strand_neg"], int(rng.integers(0, 80)), seed=s))
    for base in (0, 1):
        attempt("pafpseudo %d" % s, lambda: pc.check_pafpseudo(eng, b, base))
        attempt("pafpseudo v0 %d" % s, lambda: pc.check_pafpseudo(eng, b, base, variant=0))
    print("seed", s, "done at %.0f s, fails %d" % (time.time() - t0, fails)); sys.stdout.flush()
print("END", k, "seeds", fails, "fails")
