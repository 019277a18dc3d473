"""Kernel-logic parity on the CPU: the HIP kernel source (wga_kernels*.h) compiled against the
SIMT emulator (tests/emu) and checked against the oracle.  These run under `-m "not gpu"`; the
same checks run on the real GPU in test_gpu_parity.py."""
import numpy as np
import pytest

import parity_cases as pc
from wgatools_amd import engine, synth


@pytest.mark.parametrize("seed,n,mean,use_m", [(1, 12, 700, False), (2, 60, 40, False),
                                               (3, 400, 3, True), (4, 2, 6000, False)])
def test_stat_random(emu, seed, n, mean, use_m):
    b = synth.make_paf_batch(seed, n, mean, 60000, use_m=use_m)
    pc.check_stat(emu, b)


def test_stat_bad_ops_and_continuations(emu):
    """N/S/H/P/other ops are CigarOpInvalid for stat; split I/D count one event"""
    M, I, D, N, S, H, P, EQ, X = range(9)
    def op(l, c): return (l << 4) | c
    recs = [
        [op(5, EQ), op(3, N), op(2, I)],                       # bad at 1
        [op(5, M), op(2, I), op(7, 9), op(1, D), op(4, 10), op(9, 10), op(3, X)],  # continuations
        [op(4, S), op(5, M)],                                  # bad at 0
        [op(1, EQ)] * 1500 + [op(2, H)] + [op(1, EQ)] * 700,   # bad op deep in a multi-tile record
        [op(2, 11)],                                           # OTHER
    ]
    ops = np.array([w for r in recs for w in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    b = dict(ops=ops, op_off=off, strand_neg=np.array([0, 1, 0, 1, 0], dtype=np.uint8))
    batch = emu.make_batch(b["ops"], b["op_off"], b["strand_neg"])
    counts, diag, _ = emu.cigar_stat(batch)
    c, d = counts.numpy(), diag.numpy()
    assert d["bad_op_idx"].tolist() == [1, int(engine.NONE), 0, 1500, 0]
    # record 1 ('-'): 2I+7cont = one inv_ins event of 9 bp; 1D+4+9 = one inv_del event of 14 bp
    assert tuple(int(x) for x in c[1]) == (5, 3, 0, 0, 0, 0, 1, 9, 1, 14, 1)


def test_stat_empty(emu):
    b = dict(ops=np.zeros(0, np.uint32), op_off=np.zeros(1, np.uint64), strand_neg=np.zeros(0, np.uint8))
    batch = emu.make_batch(b["ops"], b["op_off"], b["strand_neg"])
    emu.cigar_stat(batch)


@pytest.mark.parametrize("seed,n,mean,pool,pre,use_m", [
    (1, 12, 700, 50000, False, False), (2, 40, 60, 20000, True, False),
    (3, 300, 3, 5000, True, True), (4, 3, 5000, 200000, False, True)])
def test_paf2maf_random(emu, seed, n, mean, pool, pre, use_m):
    b = synth.make_paf_batch(seed, n, mean, pool, use_m=use_m)
    rng = np.random.default_rng(seed)
    p = (rng.integers(0, 40, n), rng.integers(0, 40, n), rng.integers(0, 5, n)) if pre else None
    pc.check_paf2maf(emu, b, pre=p)


def test_paf2maf_edge_cases(emu):
    b = pc.edge_case_batch(emu)
    n = len(b["strand_neg"])
    rng = np.random.default_rng(5)
    pc.check_paf2maf(emu, b)
    pc.check_paf2maf(emu, b, pre=(rng.integers(0, 33, n), rng.integers(0, 33, n), rng.integers(0, 3, n)))
    pc.check_paf2maf(emu, b, force_slow=1)
    pc.check_paf2maf(emu, b, no_table=1)


def test_paf2maf_no_table_random(emu):
    """binary-search event lookup (the path of tiles wider than 65536 columns)"""
    pc.check_paf2maf(emu, synth.make_paf_batch(8, 30, 500, 60000), no_table=1)


def test_paf2maf_force_slow_random(emu):
    b = synth.make_paf_batch(5, 10, 300, 20000)
    pc.check_paf2maf(emu, b, force_slow=1)


def test_paf2maf_errors(emu):
    """InvalidBase position (reverse order), CigarOpInvalid, insert_str panic"""
    cigars = ["10=", "10=", "6M1I", "3=1D", "4=2N4="]
    strands = [1, 1, 0, 0, 0]
    t = [b"ACGTACGTAC", b"ACGTACGTAC", b"ACGT", b"ACGT", b"ACGTACGT"]
    q = [b"ACGTRCGYAC", b"ACGTACGTAC", b"ACGTACG", b"AC", b"ACGTACGT"]
    b = pc.batch_from_texts(emu, cigars, strands, t, q)
    r = pc.check_paf2maf(emu, b)
    d = r["diag"]
    assert int(d["bad_base_pos"][0]) == 2          # 'Y' is hit first when scanning from the end
    assert int(d["bad_base_pos"][1]) == int(engine.NONE)
    assert int(d["panic_op_idx"][2]) == 1 and int(d["panic_op_idx"][3]) == 1
    assert int(d["bad_op_idx"][4]) == 1


def test_paf2maf_long_record_many_tiles(emu):
    """one record over ~300 tiles (walk-back over > 256 tile summaries) next to short ones"""
    b = synth.make_paf_batch(11, 5, 4, 1_500_000)
    big = synth.make_paf_batch(12, 1, 300_000, 1_500_000, sigma=0.01)
    # splice the big record in the middle
    k = 2
    ops = np.concatenate([b["ops"][:int(b["op_off"][k])], big["ops"], b["ops"][int(b["op_off"][k]):]])
    lens = np.diff(b["op_off"]).astype(np.int64).tolist()
    lens.insert(k, len(big["ops"]))
    off = np.cumsum([0] + lens).astype(np.uint64)
    def ins(a, v): return np.insert(a, k, v)
    nb = dict(ops=ops, op_off=off, strand_neg=ins(b["strand_neg"], 1), t_pool=big["t_pool"],
              q_pool=big["q_pool"],
              t_src_off=ins(b["t_src_off"], big["t_src_off"][0]), t_src_len=ins(b["t_src_len"], big["t_src_len"][0]),
              q_src_off=ins(b["q_src_off"], big["q_src_off"][0]), q_src_len=ins(b["q_src_len"], big["q_src_len"][0]))
    pc.check_stat(emu, nb)
    pc.check_paf2maf(emu, nb)


def test_paf2maf_v1_kernel(emu):
    """the battery of the row kernels on v1 (expand_variant 0: the kernel of the tiles the streaming kernel leaves)"""
    pc.window_kernel_cases(emu, variant=0)


def test_paf2maf_row_kernels_dense_indels(emu):
    b = pc.dense_indel_batch(emu)
    pc.check_paf2maf(emu, b, variant=3)
    pc.check_paf2maf(emu, b, variant=0)


def test_paf2maf_row_kernels_errors_and_long_record(emu):
    cigars = ["10=", "10=", "6M1I", "3=1D", "4=2N4="]
    strands = [1, 1, 0, 0, 0]
    t = [b"ACGTACGTAC", b"ACGTACGTAC", b"ACGT", b"ACGT", b"ACGTACGT"]
    q = [b"ACGTRCGYAC", b"ACGTACGTAC", b"ACGTACG", b"AC", b"ACGTACGT"]
    r = pc.check_paf2maf(emu, pc.batch_from_texts(emu, cigars, strands, t, q), variant=3)
    d = r["diag"]
    assert int(d["bad_base_pos"][0]) == 2 and int(d["bad_base_pos"][1]) == int(engine.NONE)
    assert int(d["panic_op_idx"][2]) == 1 and int(d["panic_op_idx"][3]) == 1
    assert int(d["bad_op_idx"][4]) == 1
    big = synth.make_paf_batch(12, 1, 40_000, 400_000, sigma=0.01)  # one record over ~40 tiles
    pc.check_paf2maf(emu, big, variant=3)
    # an invalid base in the middle of a long '-' strand record whose slice overlaps another record's
    bad = synth.make_paf_batch(13, 2, 3000, 100_000)
    bad["strand_neg"][:] = 1
    qp = bad["q_pool"].copy()
    k = int(bad["q_src_off"][1] + bad["q_src_len"][1] // 2)
    qp[k] = ord("R")
    bad["q_pool"] = qp
    pc.check_paf2maf(emu, bad, variant=3)


def test_paf2maf_stream_kernel(emu):
    pc.window_kernel_cases(emu, variant=3)


def test_paf2maf_stream_kernel_jobs_and_skips(emu):
    pc.stream_kernel_cases(emu)


def test_expand_variant_default(emu):
    """expand_variant -1 (the default) is the streaming kernel whatever the batch (the window kernel that took records of a few
    dozen ops was retired in round 6); 2 is refused; the bytes are the oracle's"""
    emu.set_param("expand_variant", -1)
    for b in (synth.make_paf_batch(41, 60, 40, 30000), synth.make_paf_batch(41, 30, 200, 30000), synth.make_paf_batch(42, 3, 3000, 90000)):
        pc.check_paf2maf(emu, b)
        assert emu.get_param("expand_variant_used") == 3
    with pytest.raises(Exception):
        emu.set_param("expand_variant", 2)
    emu.set_param("expand_variant", pc.DEFAULT_EXPAND_VARIANT)


def test_paf2maf_maf2paf_roundtrip(emu):
    assert pc.check_paf2maf_maf2paf_roundtrip(emu, 5, 24, 180) > 3000


def test_scan_and_scatter(emu):
    rng = np.random.default_rng(3)
    for n in (0, 1, 5, 1024, 1025, 5000, 300000):
        v = rng.integers(0, 1 << 40, n).astype(np.uint64)
        got = emu.exclusive_scan_u64(n, emu.upload(v) if n else None).numpy()
        exp = np.concatenate([np.zeros(1, np.uint64), np.cumsum(v, dtype=np.uint64)])
        assert (got == exp).all()
    lens = rng.integers(0, 70, 50)
    src_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    src = rng.integers(0, 256, int(src_off[-1])).astype(np.uint8)
    dst_off = (np.arange(50) * 100).astype(np.uint64)
    dst = emu.empty(5000, np.uint8).fill(0)
    emu.scatter_bytes(50, emu.upload(src), emu.upload(src_off), dst, emu.upload(dst_off))
    d = dst.numpy()
    for i in range(50):
        assert (d[i * 100:i * 100 + lens[i]] == src[int(src_off[i]):int(src_off[i + 1])]).all()
        assert (d[i * 100 + lens[i]:(i + 1) * 100] == 0).all()


def _cov_problem(seed, n, mean, nt):
    rng = np.random.default_rng(seed)
    b = pc.sprinkle_ops(rng, synth.make_paf_batch(seed, n, mean, 1000))
    span = synth.class_sums(b["ops"] & 15, b["ops"] >> 4, b["op_off"])
    tlen = rng.integers(200, 3000, nt)
    tid = rng.integers(0, nt, n)
    # some records run off the end of their target (positions >= length are ignored)
    tstart = (rng.random(n) * tlen[tid] * 1.05).astype(np.uint64)
    return b, tid, tstart, tlen


@pytest.mark.parametrize("seed,n,mean,nt,align", [(1, 30, 40, 3, 4), (2, 200, 5, 7, 1), (3, 4, 3000, 2, 4)])
def test_pafcov(emu, seed, n, mean, nt, align):
    b, tid, tstart, tlen = _cov_problem(seed, n, mean, nt)
    pc.check_pafcov(emu, b, tid, tstart, tlen, align=align)
    pc.check_pafcov(emu, b, tid, tstart, tlen, align=align, split=True)


def test_pafcov_long_target(emu):
    """a target longer than several scan chunks, with a record spanning tiles"""
    b = synth.make_paf_batch(9, 3, 2500, 1000)
    pc.check_pafcov(emu, b, [0, 0, 1], [10, 5000, 0], [40000, 9000])


def test_pafcov_ops_across_many_windows(emu):
    pc.check_pafcov_long_ops(emu)


def test_pafcov_segments_across_hundreds_of_windows(emu):
    """K5's list pass takes a lane per window, 64 windows a round: ops of 2 000 000 / 1 500 000 / 900 000 bases (segments across
    244 / 183 / 110 windows: several rounds, the pieces beyond the tile's own slots in the list regions, which grow)"""
    mk = lambda p: [(int(ln) << 4) | int(c) for c, ln in p]
    recs = [mk([(7, 5), (7, 2_000_000), (2, 3), (0, 1_500_000), (8, 1), (7, 700_000)]),
            mk([(7, 3), (1, 2)] * 300 + [(7, 900_000)] + [(8, 1), (7, 2)] * 200),
            mk([(7, 1)] * 40)]
    ops = np.array([o for r in recs for o in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    b = dict(ops=ops, op_off=off, strand_neg=np.zeros(3, dtype=np.uint8))
    pc.check_pafcov(emu, b, [0, 1, 0], [100, 8191, 4_300_000], [4_400_000, 1_000_000])


def test_pafcov_many_small_targets(emu):
    pc.check_pafcov_many_small_targets(emu, nt=120)


def test_pafcov_random_shapes(emu):
    pc.check_pafcov_random(emu, 11, 5)


def test_pafcov_look_back(emu, monkeypatch):
    """K5's list pass: a tile's first segment takes its record's position from the sums the tiles in front of it published —
    records across 2 .. 70 tiles (more than one round of 64 lanes), tiles that end exactly with a record, and the same with the
    look-back told to add up the ops itself (what it does when a tile in front has not published in time)"""
    pc.check_pafcov_look_back(emu)
    keep = emu.get_param("cov_spin_limit")
    emu.set_param("cov_spin_limit", 0)
    try:
        pc.check_pafcov_look_back(emu)
    finally:
        emu.set_param("cov_spin_limit", keep)


@pytest.mark.parametrize("base", [0, 1])
@pytest.mark.parametrize("seed,n,mean", [(1, 20, 60), (2, 150, 4), (3, 3, 2600)])
def test_pafpseudo(emu, base, seed, n, mean):
    rng = np.random.default_rng(seed)
    b = synth.make_paf_batch(seed, n, mean, 60000)
    b = pc.sprinkle_ops(rng, b, codes=(3, 5, 6, 11))      # N H P other: ignored
    # S consumes query like I: give those records a longer slice
    ops = b["ops"].copy()
    k = rng.integers(0, len(ops), max(1, len(ops) // 50))
    ops[k] = (ops[k] & ~np.uint32(15)) | np.uint32(4)
    b["ops"] = ops
    code, length = ops & 15, (ops >> 4).astype(np.uint64)
    v = np.where((code == 0) | (code == 7) | (code == 8) | (code == 1) | (code == 4), length, 0).astype(np.uint64)
    c = np.concatenate([np.zeros(1, np.uint64), np.cumsum(v, dtype=np.uint64)])
    b["q_src_len"] = c[b["op_off"][1:].astype(np.int64)] - c[b["op_off"][:-1].astype(np.int64)]
    b["q_src_off"] = (rng.random(n) * (len(b["q_pool"]) - b["q_src_len"].astype(np.float64))).astype(np.uint64)
    skip = np.where(rng.random(n) < 0.4, rng.integers(0, 30, n), 0)
    pc.check_pafpseudo(emu, b, base, skip=None)
    # trimmed heads (pseudomaf.rs:190-192) — keep skip below the segment length
    seg = synth.class_sums(code, ops >> 4, b["op_off"])
    skip = np.minimum(skip, (seg["mx"] + seg["d"]).astype(np.int64))
    pc.check_pafpseudo(emu, b, base, skip=skip)


def test_pafpseudo_stream_kernel(emu):
    pc.pseudo_stream_cases(emu)


def test_pafpseudo_symbol_runs(emu):
    """symbol mode, per-granule walk over the ops that cover it: long X / D runs, rows that start inside a granule, dense
    single-column ops, one op of 1.2 M columns, trimmed heads up to 8191 columns"""
    cigars = ["100=50X9000D3=1X40D8000=33X7=", "5=1200000D5=", "1=1X" * 600 + "4=", "70D", "70X", "3=2I4=1X5=",
              "20=" + "35X2=" * 300, "9000=1D9000=1X100="]
    strands = [0, 1, 0, 0, 1, 0, 0, 1]
    b = pc.batch_from_texts(emu, cigars, strands, [b"A"] * 8, [b"A"] * 8)
    pc.check_pafpseudo(emu, b, 0)
    pc.check_pafpseudo(emu, b, 0, skip=[0, 3, 17, 0, 5, 2, 33, 8191])


def test_pafpseudo_length_mismatch(emu):
    """slice longer (tail kept) / shorter (panic) than the CIGAR consumes"""
    cigars = ["5=2I3=", "5=2I3=", "4=3D4=", "10=", "8=2I", "8=1D", "3=2S1="]
    strands = [0, 1, 1, 0, 0, 1, 0]
    q = [b"ACGTACGTACGTTT", b"ACGTACGTACGTTT", b"ACGTAC", b"ACGTA", b"ACGTA", b"ACGTA", b"ACGTA"]
    t = [b"A"] * 7
    b = pc.batch_from_texts(emu, cigars, strands, t, q)
    pc.check_pafpseudo(emu, b, 1)


def test_maf_pair_stat(emu):
    rng = np.random.default_rng(4)
    pairs, strands = [], []
    # rows produced by paf2maf itself ...
    b = synth.make_paf_batch(21, 12, 120, 30000)
    for i in range(12):
        try:
            pairs.append(pc.oracle_rows(b, i))
            strands.append(int(b["strand_neg"][i]))
        except Exception:
            pass
    # ... and free-form rows: '-'/'-' columns count as '=', case matters, unequal lengths zip
    for L in (0, 1, 15, 16, 17, 63, 64, 65, 200, 1000, 1023, 1024, 1025, 1041, 2100):
        t = pc.rand_seq(rng, L, b"ACGTacgt--N")
        q = pc.rand_seq(rng, L + int(rng.integers(0, 3)), b"ACGTacgt--N")
        pairs.append((t, q))
        strands.append(L & 1)
    from helpers import GOLDEN, read_maf_blocks
    import os
    blk = read_maf_blocks(os.path.join(GOLDEN, "test.maf"))[0]
    pairs.append((blk[0]["seq"], blk[1]["seq"]))
    strands.append(0)
    pairs += pc.binary_row_pairs(rng)
    strands += [0, 1, 0, 1]
    pc.check_maf_pair(emu, pairs, strands)
    pc.check_maf_call_runs(emu, pairs)


def test_maf_walks_pairs_of_every_shape(emu):
    """the walks take up to eight consecutive blocks of a wave as ONE column stream (`maf_group`; a lane holds 32 columns of one
    block, a block's last lane its last 32 bytes): every pair of lengths around the 32-column and the 2 048-column borders
    (a block ends on a step's border, inside it, in its last lane; empty and tiny blocks in between), groups of 1, 3 and 8,
    counters and run lists of both walks"""
    rng = np.random.default_rng(11)
    lens = [0, 1, 31, 32, 33, 63, 64, 65, 500, 2016, 2047, 2048, 2049, 2080, 4095, 4096, 4097]
    pairs, strands = [], []
    for la in lens:
        for lb in lens:
            for L in (la, lb):
                t = pc.rand_seq(rng, L, b"ACGTacgt--N")
                q = bytearray(pc.rand_seq(rng, L, b"ACGTacgt--N"))
                for k in range(0, L, 3):                       # mostly equal columns, as in real blocks: longer runs
                    if rng.random() < 0.8:
                        q[k:k + 3] = t[k:k + 3]
                pairs.append((t, bytes(q)))
                strands.append(int(rng.integers(0, 2)))
    try:
        for g in (1, 3, 8):
            emu.set_param("maf_group", g)
            pc.check_maf_pair(emu, pairs[g::2] if g == 3 else pairs, strands[g::2] if g == 3 else strands)
            pc.check_maf_call_runs(emu, pairs[g::2] if g == 3 else pairs)
    finally:
        emu.set_param("maf_group", 0)


def test_maf_stream_groups_with_long_and_binary_blocks(emu):
    """a group of eight whose blocks are of every kind at once — long (left to the piece walk, whose packed totals are folded
    every three steps in this build), tiny, empty, rows that are not text — and a long block whose last piece is tiny"""
    rng = np.random.default_rng(12)
    pairs, strands = [], []
    for L in (7000, 5, 0, 300, 64 * 100 + 3, 40, 31, 9000, 2048, 1, 33, 6500):
        t = pc.rand_seq(rng, L, b"ACGTacgt--N")
        q = pc.rand_seq(rng, L + int(rng.integers(0, 3)), b"ACGTacgt--N")
        pairs.append((t, q))
        strands.append(L & 1)
    pairs += pc.binary_row_pairs(rng)
    strands += [0, 1, 0, 1]
    try:
        for g, long_cols, piece_cols in ((8, 6000, 64), (8, 6000, 2048), (5, 32768, 16384), (8, 300, 100), (8, 3000, 7000)):
            emu.set_param("maf_group", g)
            emu.set_param("maf_long_cols", long_cols)
            emu.set_param("maf_piece_cols", piece_cols)
            pc.check_maf_pair(emu, pairs, strands)
            pc.check_maf_call_runs(emu, pairs)
    finally:
        emu.set_param("maf_group", 0)
        emu.set_param("maf_long_cols", 32768)
        emu.set_param("maf_piece_cols", 16384)


def test_pafpseudo_stream_random(emu):
    """pafpseudo's rows through the streaming kernel on random CIGARs with clips, runs of insertions and clips on one column,
    zero-length ops and random trimmed heads, job sizes 1 and 8, both modes — against the oracle"""
    before = emu.get_param("expand_job_tiles")
    try:
        for seed in range(12):
            rng = np.random.default_rng(1000 + seed)
            emu.set_param("expand_job_tiles", 1 if seed % 2 else 8)
            cigars, strands = [], []
            for _ in range(int(rng.integers(2, 7))):
                parts = ["%d%s" % (rng.integers(0, 30), "SI"[int(rng.integers(0, 2))])] if rng.random() < 0.5 else []
                for _ in range(int(rng.integers(1, 700))):
                    r = rng.random()
                    if r < 0.55:
                        parts.append("%d%s" % (rng.integers(1, 60), "=XM"[int(rng.integers(0, 3))]))
                    elif r < 0.70:
                        parts.append("%dD" % rng.integers(0, 40))
                    elif r < 0.85:
                        parts.append("%dI" % rng.integers(0, 40))
                    elif r < 0.90:
                        parts.append("%dS" % rng.integers(0, 9))
                    elif r < 0.95:
                        parts.append("".join("%d%s" % (rng.integers(1, 4), "IS"[int(rng.integers(0, 2))]) for _ in range(int(rng.integers(2, 400)))))
                    else:
                        parts.append("%d%s" % (rng.integers(1, 5), "NHP"[int(rng.integers(0, 3))]))
                parts.append("%d=" % rng.integers(1, 2000))
                cigars.append("".join(parts))
                strands.append(int(rng.integers(0, 2)))
            qs = [pc.rand_seq(rng, pc.pseudo_consumption(c)) for c in cigars]
            b = pc.batch_from_texts(emu, cigars, strands, [b"A"] * len(cigars), qs, pad=int(rng.integers(40, 90)))
            b["q_pool"] = np.concatenate([np.frombuffer(b"N" * 64, np.uint8), b["q_pool"], np.frombuffer(b"N" * 64, np.uint8)])
            b["q_src_off"] = b["q_src_off"] + np.uint64(64)
            cols = [sum(int(n) for n, op in __import__("re").findall(r"(\d+)(\D)", c) if op in "M=XD") for c in cigars]
            skip = [int(rng.integers(0, c + 1)) if rng.random() < 0.6 else 0 for c in cols]
            for mode in (1, 0):
                pc.check_pafpseudo(emu, b, mode, skip=skip, variant=3)
                assert emu.get_param("pseudo_stream_left_to_blocks") == 0
    finally:
        emu.set_param("expand_job_tiles", before)


def test_fast_expected_matches_oracle():
    """the numpy expectation used for long records on the GPU agrees with the C oracle"""
    b = synth.make_paf_batch(31, 25, 150, 40000)
    for i in range(25):
        t = b["t_pool"][int(b["t_src_off"][i]):int(b["t_src_off"][i] + b["t_src_len"][i])].tobytes()
        q = b["q_pool"][int(b["q_src_off"][i]):int(b["q_src_off"][i] + b["q_src_len"][i])].tobytes()
        assert pc.fast_expected_rows(pc.rec_ops(b, i), t, q, b["strand_neg"][i]) == pc.oracle_rows(b, i)


def test_paf_call_events(emu):
    b = synth.make_paf_batch(33, 12, 150, 400000)
    for svlen, snp in ((0, True), (3, False), (50, True)):
        assert pc.check_paf_call_events(emu, b["ops"], b["op_off"], svlen, snp) > 0
    # invalid op mid-record, zero-length ops, lengths split by the packer (head + continuation pieces)
    L = (1 << 28) - 1
    mk = lambda *p: [(ln << 4) | c for c, ln in p]
    recs = [mk((7, 5), (1, 3), (3, 9), (8, 1), (2, 7)),                       # N stops the walk
            mk((7, 5), (1, L), (9, L), (9, 12), (7, 2), (2, 1), (10, 0), (8, 3)),  # split I, split D of small total
            mk((1, 9), (7, 1), (2, 9), (1, 9), (0, 0), (1, 9)),                # leading I, I after D, 0M then I
            mk((8, 2), (2, L), (10, L), (8, 1)),
            [], mk((11, 4), (7, 3))]
    ops = np.array([o for r in recs for o in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    for svlen, snp in ((0, True), (8, True), (1 << 40, False)):
        pc.check_paf_call_events(emu, ops, off, svlen, snp)


def test_paf_call_long_records_in_pieces(emu):
    pc.check_paf_call_long_records(emu, mops=0)
    ops, off = pc.long_record_ops(7, 1, 40_000)     # the product's piece size: 5 pieces (2 Mop records run on the GPU)
    pc.check_paf_call_events(emu, ops, off, 20, True)


def test_dotplot_long_records_in_pieces(emu):
    pc.check_dotplot_long_records(emu, mops=0)


def test_piece_table_kept_or_rebuilt(emu):
    pc.check_piece_table_rebuild(emu)


def test_cigar_chain_long_records_in_pieces(emu):
    pc.check_cigar_chain_long_records(emu, mops=0)
    ops, off = pc.long_record_ops(7, 1, 40_000)     # the product's piece size
    pc.check_cigar_chain(emu, ops, off)


def test_device_tokeniser(emu):
    pc.check_tokeniser(emu, pc.TOKENISER_EDGE_TEXTS)
    b = synth.make_paf_batch(41, 10, 300, 300000)
    texts = [synth.cigar_text(pc.rec_ops(b, i)).encode() for i in range(10)]
    pc.check_tokeniser(emu, texts + [b"3M", b""] + texts[:3])


def test_pafcov_format(emu):
    rng = np.random.default_rng(3)
    pc.check_pafcov_format(emu, b"chr1", [0, 1, 9, 10, 99, 100, 2147483647, 12345], 0)
    pc.check_pafcov_format(emu, b"g01#1#chr1", rng.integers(0, 500, 1300), 95)
    pc.check_pafcov_format(emu, b"t", rng.integers(0, 3, 40), 999_999_990)
    pc.check_pafcov_format(emu, b"big", rng.integers(0, 70000, 30), 9_999_999_990)
    pc.check_pafcov_format(emu, b"huge", [7, 8], 18_446_744_073_709_551_000)
    pc.check_pafcov_format(emu, b"", [5], 41)
    pc.check_pafcov_format(emu, b"none", [], 0)
    pc.check_pafcov_format(emu, b"a_target_name_longer_than_the_staging_buffer_takes_512_lines_of", rng.integers(0, 500, 1100), 7)
    pc.check_pafcov_format(emu, b"c", rng.integers(0, 9, 512 * 3), 999_999_000)   # exactly three blocks, a digit roll-over inside


def test_device_tokeniser_random_bytes(emu):
    """adversarial texts: digits, op letters and stray bytes in any order — the device tokeniser must agree
    with the host packer on ops, error code and error token for every one of them"""
    from hypothesis import given, settings, strategies as st, HealthCheck
    alphabet = [bytes([c]) for c in b"0123456789"] * 3 + [bytes([c]) for c in b"MIDNSHP=XB"] * 2 + \
               [b" ", b"\t", b"\xc3\xa9", b"\xe2\x82\xac", b"\xff", b"0000000000", b"99999999", b"268435455", b"268435456"]
    texts_st = st.lists(st.lists(st.sampled_from(alphabet), min_size=0, max_size=40).map(b"".join), min_size=1, max_size=12)

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(texts_st)
    def run(texts):
        # keep split lengths bounded: a length of 10^17 would pack to 4·10^8 ops (u64 overflows are fine:
        # they are errors and pack to nothing)
        import re
        texts = [t for t in texts if all(int(r) < (1 << 33) or int(r) > 0xFFFFFFFFFFFFFFFF for r in re.findall(rb"[0-9]+", t))]
        if texts:
            pc.check_tokeniser(emu, texts)
    run()


def test_paf2maf_drain_min_settings(emu):
    pc.check_drain_min_settings(emu, synth.make_paf_batch(21, 6, 900, 200_000))


def test_cigar_chain(emu):
    b = synth.make_paf_batch(57, 12, 300, 400000)
    pc.check_cigar_chain(emu, b["ops"], b["op_off"])
    L = (1 << 28) - 1
    mk = lambda *p: [(ln << 4) | c for c, ln in p]
    recs = [mk((1, 2), (2, 3), (7, 5), (1, 2), (7, 3), (8, 1), (2, 4), (1, 2), (2, 1)),     # head + tail indels
            mk((7, 9)), mk((1, 4)), mk((2, 4), (1, 1)), mk((0, 0), (1, 3), (0, 5)),          # M only, I only, no M, 0M
            mk((7, 5), (1, 3), (3, 9), (7, 1)), mk((3, 2), (7, 4)),                            # N stops the fold
            mk((7, 5), (1, L), (9, L), (9, 12), (7, 2), (2, L), (10, 5)),                     # split I / D, split D in the tail
            mk((7, 3), (1, 1), (2, 2), (1, 3), (8, 1), (2, 2), (2, 3), (7, 7)),               # mixed groups, repeated kinds
            mk(*([(7, 1), (1, 1)] * 300 + [(7, 2)])), mk(*([(1, 1)] * 260 + [(7, 1)] + [(2, 2)] * 270))]  # > 256 ops
    ops = np.array([o for r in recs for o in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    pc.check_cigar_chain(emu, ops, off)


def test_cigar_chain_steps(emu):
    for seed in (1, 2):
        ops, off = pc.chain_stress_records(seed)
        pc.check_cigar_chain(emu, ops, off)


def test_runs_bridge_synthetic(emu):
    L = (1 << 28) - 1
    recs = [[(5, 0), (3, 1), (2, 3), (4, 2)], [], [(L, 0), (L + 1, 1), (2 * L + 7, 2), (1, 3), (3 * L, 3)],
            [(1, c) for c in (0, 1, 2, 3)] * 200, [(10 ** 9, 2)], [(123456789, 0)]]
    pc.check_runs_bridge_synthetic(emu, recs)


def test_pafpseudo_fill_without_the_class_sums_call(emu):
    """wga_pafpseudo_fill takes the sums wga_cigar_class_sums left for the same batch — or computes them when no such call
    stands in front (other arrays, or none at all)"""
    b = synth.make_paf_batch(7, 40, 300, 60000)
    code, length = b["ops"] & 15, (b["ops"] >> 4).astype(np.uint64)
    v = np.where((code == 0) | (code == 7) | (code == 8) | (code == 1) | (code == 4), length, 0).astype(np.uint64)
    c = np.concatenate([np.zeros(1, np.uint64), np.cumsum(v, dtype=np.uint64)])
    b["q_src_len"] = c[b["op_off"][1:].astype(np.int64)] - c[b["op_off"][:-1].astype(np.int64)]
    b["q_src_off"] = np.zeros(40, dtype=np.uint64)
    for base in (0, 1):
        pc.check_pafpseudo(emu, b, base, sums_call=False)
        pc.check_pafpseudo(emu, b, base)
        pc.check_pafpseudo(emu, b, base, sums_call=False)


def test_elem_scan_reuse(emu):
    pc.check_elem_scan_reuse(emu)


def test_bridge_blocks(emu):
    pc.check_bridge_blocks(emu)


def test_chain_lines(emu):
    rng = np.random.default_rng(77)
    L = (1 << 28) - 1
    recs = [[(10, 2, 0), (5, 0, 3), (7, 1, 1), (4, 0, 0)], [(9, 0, 0)], [],
            [(0, 0, 4), (3, 0, 0), (0, 2, 0), (6, 0, 0)],              # zero sizes: "0M" in the text, no op
            [(L + 5, 2 * L + 1, L), (1, 0, 0)],                         # lengths beyond one packed op
            [(int(rng.integers(1, 50)), int(rng.integers(0, 4)), int(rng.integers(0, 4))) for _ in range(700)] + [(3, 0, 0)]]
    strands = [0, 1, 0, 1, 0, 1]
    pc.check_chain_lines(emu, recs, strands)
    # rows: sequences exactly as long as the lines consume, one too short (insert_str panics), one longer (tail copied)
    recs2, seqs, strands2 = [], [], []
    for k in range(9):
        r = [(int(rng.integers(1, 40)), int(rng.integers(0, 3)) * int(rng.integers(0, 9)),
              int(rng.integers(0, 3)) * int(rng.integers(0, 9))) for _ in range(int(rng.integers(1, 90)))]
        r[-1] = (r[-1][0], 0, 0)
        tn = sum(s + qd for s, qd, td in r)
        qn = sum(s + td for s, qd, td in r)
        extra = (0, 0, 5, -3, 0, 2, 0, 0, -200)[k]
        recs2.append(r)
        seqs.append((pc.rand_seq(rng, max(tn + extra, 0), b"ACGTN"), pc.rand_seq(rng, max(qn + extra, 0), b"ACGTNacgtn")))
        strands2.append(k & 1)
    pc.check_chain_lines(emu, recs2, strands2, seqs=seqs)


def test_dotplot_segments(emu):
    b = synth.make_paf_batch(91, 14, 700, 500000)
    pc.sprinkle_ops(np.random.default_rng(2), b, frac=0.03)
    for cutoff in (0, 3, 50, 10 ** 6):
        pc.check_dotplot(emu, b["ops"], b["op_off"], b["strand_neg"], cutoff)
    L = (1 << 28) - 1
    mk = lambda *p: [(ln << 4) | c for c, ln in p]
    recs = [mk((1, 9), (2, 3), (7, 5), (1, 2), (7, 3), (8, 1), (2, 40), (1, 2), (2, 1)),    # leading indels, small + long
            mk((7, 9)), mk((1, 40)), mk((2, 4), (1, 1)), mk((0, 0), (1, 30), (0, 0), (2, 0), (7, 5)),   # zero lengths
            mk((7, 5), (1, 3), (3, 9), (4, 2), (7, 1), (11, 6), (2, 99), (5, 1)),              # ignored ops keep the segment open
            mk((7, 5), (1, L), (9, L), (9, 12), (7, 2), (2, L), (10, 5), (7, 1)),              # split I / D: serial walk
            mk(*([(7, 1), (1, 30), (2, 1)] * 300 + [(7, 2)])), mk(*([(1, 1)] * 260 + [(7, 1)] + [(2, 20)] * 270)),  # > 256 ops
            mk(*([(7, 3), (1, 2)] * 129)), []]
    ops = np.array([o for r in recs for o in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    strands = [k & 1 for k in range(len(recs))]
    for cutoff in (0, 10, 2 * L):
        pc.check_dotplot(emu, ops, off, strands, cutoff)
    rng = np.random.default_rng(12)
    pairs, st = [], []
    for L2 in (0, 1, 17, 64, 300, 1025, 2100):
        pairs.append((pc.rand_seq(rng, L2, b"ACGTacgt--N"), pc.rand_seq(rng, L2 + int(rng.integers(0, 3)), b"ACGTacgt--N")))
        st.append(L2 & 1)
    for cutoff in (0, 1, 5):
        pc.check_dotplot_maf(emu, pairs, st, cutoff)


def test_paf_split(emu):
    rng = np.random.default_rng(3)
    b = synth.make_paf_batch(23, 40, 600, 200000)
    rows = []
    for i in range(40):
        cg = pc.rec_text(b, i)
        tags = ["NM:i:%d" % i, "tp:A:P", cg, "zd:i:3"][: 2 + int(rng.integers(0, 3))]
        if i % 7 == 3:
            tags = ["cg:Z:5=", cg]                     # the first cg:Z: wins
        if i % 11 == 5:
            tags = ["NM:i:0"]                          # no CIGAR tag: not a parse error
        rows.append("q%d\t%d\t%d\t%d\t%s\tchr%d\t%d\t%d\t%d\t%d\t%d\t%d\t%s" % (
            i, 10 ** 9 + i, i, i + 5, "-+"[i & 1], i % 3, 18446744073709551615 if i == 2 else 2 * 10 ** 9, 7 * i, 7 * i + 9, i,
            2 * i, 60, "\t".join(tags)))
    clean = ("# header comment\n" + "\n".join(rows[:20]) + "\n\n# mid\tcomment \"x\"\n" + "\n".join(rows[20:])).encode()
    pc.check_paf_split(emu, clean + b"\n")
    pc.check_paf_split(emu, clean)                       # no newline at the end
    pc.check_paf_split(emu, b"")
    pc.check_paf_split(emu, b"\n\n")
    odd = [b"q\t1\t2\t3\t+\tt\t4\t5\t6\t7\t8",                      # 11 fields
           b"q\t1\t2\t3\t*\tt\t4\t5\t6\t7\t8\t9\tcg:Z:5=",          # strand
           b"q\t1\t2x\t3\t+\tt\t4\t5\t6\t7\t8\t9\tcg:Z:5=",         # digit
           b"q\t\t2\t3\t+\tt\t4\t5\t6\t7\t8\t9\tcg:Z:5=",           # empty integer
           b"q\t18446744073709551616\t2\t3\t+\tt\t4\t5\t6\t7\t8\t9",    # overflow
           b"q\t+1\t2\t3\t+\tt\t4\t5\t6\t7\t8\t9\tcs:Z::5",          # '+1' parses; cs instead of cg
           b"\"q\"\t1\t2\t3\t+\tt\t4\t5\t6\t7\t8\t9\tcg:Z:5=",       # quoted field
           b"q\t1\t2\t3\t+\tt\t4\t5\t6\t7\t8\t9\tcg:Z:5=\r",         # CRLF
           b"q\t1\t2\t3\t+\tt\t4\t5\t6\t7\t8\t9\tcs:Z::5\tcg:Z:7=",   # both: cg wins
           b"#c\rq\t1", b"q\t1\t2\t3\t-\tt\t4\t5\t6\t7\t8\t9"]
    pc.check_paf_split(emu, b"\n".join(odd) + b"\n" + clean)
    # delimiters right at the 16-byte / 4096-byte block edges
    for pad in (4080, 4095, 4096, 4097, 8191):
        pc.check_paf_split(emu, b"#" + b"x" * (pad - 1) + b"\n" + rows[0].encode() + b"\n" + b"\t" * 40 + b"\n")


def test_maf_split(emu):
    rng = np.random.default_rng(8)
    def block(k, cols, extra=b""):
        t = pc.rand_seq(rng, cols, b"ACGTacgt-N")
        q = pc.rand_seq(rng, cols, b"ACGTacgt-N")
        return (b"a score=%d\n" % k + b"s ref.chr%d   %d %d + 1000000 " % (k, 7 * k, cols) + t + b"\n" +
                b"s\tqry.%d\t%d\t%d\t-\t+2000000\t" % (k, 11 * k, cols) + q + extra + b"\n\n")
    clean = b"##maf version=1 scoring=x\n# a comment\n" + b"".join(block(k, c) for k, c in enumerate((1, 15, 16, 17, 300, 4100, 9000)))
    pc.check_maf_split(emu, clean)
    pc.check_maf_split(emu, clean[:-2])                      # no newline at the end
    pc.check_maf_split(emu, b"")
    pc.check_maf_split(emu, b"s first line is the header even if it looks like an s-line\ns a 1 2 + 3 ACGT\n")
    odd = [b"s a 1 2 + 3", b"s a 1 2 + 3 ACGT extra", b"s a x 2 + 3 ACGT", b"s a 1 2 * 3 ACGT", b"s a 1 2 + 18446744073709551616 ACGT",
           b"sX a 1 2 + 3 ACGT", b" s a 1 2 + 3 ACGT", b"s a 1 2 + 3 ACGT \r", b"s\x0ba\x0c1 2 + 3 ACGT", b"s a\xc2\xa01 2 + 3 ACGT",
           b"i a N 0 C 0", b"e a 1 2 + 3 I", b"q a 99", b"", b"s", b"s a +1 2 - 3 AC-GT"]
    pc.check_maf_split(emu, b"##maf\n" + b"\n".join(odd) + b"\n" + clean)
    pc.check_maf_split(emu, (b"##maf\n" + b"\n".join(odd)).replace(b"\n", b"\r\n"))
    for pad in (4079, 4095, 4096, 4097):                     # delimiters at the block edges
        pc.check_maf_split(emu, b"#" + b"x" * (pad - 1) + b"\n" + block(3, 50) + b" " * 40 + b"\n")






def test_fasta_pool(emu):
    for t in pc.FASTA_CASES:
        pc.check_fasta_pool(emu, t)
    rng = np.random.default_rng(3)
    for k in range(6):
        pc.check_fasta_pool(emu, pc.random_fasta(rng, int(rng.integers(1, 9)), 9000, crlf=bool(k & 1)))
    pc.check_fasta_pool(emu, pc.random_fasta(rng, 3, 30000, width=60))      # the usual 60-column layout, several blocks


def test_maf_long_blocks_piecewise(emu):
    """blocks beyond `maf_long_cols` are walked in pieces of `maf_piece_cols` (one wave each, class of the column in front
    carried in, run slots and non-gap prefixes from a scan over the pieces): same counters and run lists, against the oracle"""
    rng = np.random.default_rng(77)
    pairs, strands = [], []
    for L in (5, 63, 64, 65, 130, 999, 1024, 2100, 5000):
        t = pc.rand_seq(rng, L, b"ACGTacgt--N")
        q = pc.rand_seq(rng, L + int(rng.integers(0, 3)), b"ACGTacgt--N")
        pairs.append((t, q))
        strands.append(L & 1)
    pairs.append((b"-" * 700 + b"ACGT" * 100, b"-" * 650 + b"A" * 50 + b"ACGA" * 100))   # long runs across piece borders
    strands.append(1)
    try:
        for long_cols, piece_cols in ((100, 64), (1, 1000), (500, 17), (64, 1024)):
            emu.set_param("maf_long_cols", long_cols)
            emu.set_param("maf_piece_cols", piece_cols)
            pc.check_maf_pair(emu, pairs, strands)
            pc.check_maf_call_runs(emu, pairs)
        # a fill call that does not find its count call's table (here: the piece size changed in between) lists and walks
        # the long blocks itself
        emu.set_param("maf_long_cols", 100)
        scan = emu.exclusive_scan_u64

        def scan_and_change(*a, **k):
            emu.set_param("maf_piece_cols", 96)
            return scan(*a, **k)
        emu.exclusive_scan_u64 = scan_and_change
        try:
            emu.set_param("maf_piece_cols", 64)
            pc.check_maf_pair(emu, pairs, strands)
            emu.set_param("maf_piece_cols", 64)
            pc.check_maf_call_runs(emu, pairs)
        finally:
            del emu.exclusive_scan_u64
    finally:
        emu.set_param("maf_long_cols", 32768)
        emu.set_param("maf_piece_cols", 16384)


def test_reduce_scatter_i32(monkeypatch):
    """wga_reduce_scatter_i32 over three emulated devices: slice g of buffer g = the sum over the devices, the rest of every
    buffer is untouched; one context is a no-op; two contexts on one device are refused.  Both forms: the peers' buffers read
    in place by one kernel per device, and ("reduce_staged") pulled into scratch first — where the emulator's streams keep the
    book of what would be in flight together on hardware: all N - 1 pulls of a device, not one after the other"""
    import ctypes as C
    from wgatools_amd import build, _lib
    monkeypatch.setenv("WGA_EMU_DEVICES", "3")
    lib = _lib.load(build.build_emu())
    engs = [engine.Engine(g, lib) for g in range(3)]
    rng = np.random.default_rng(4)
    for staged in (0, 1):
        for e in engs:
            e.set_param("reduce_staged", staged)
        for count in (0, 1, 7, 1000, 70001):
            host = [rng.integers(-1000, 1000, max(count, 1), dtype=np.int32) for _ in range(3)]
            bufs = [e.upload(h) for e, h in zip(engs, host)]
            if count == 1000:       # a buffer that starts inside a 16-byte group: the kernel's counter-by-counter form
                bufs[1] = engs[1].upload(np.concatenate([np.zeros(1, np.int32), host[1]]))
                ptrs = [bufs[0].ptr, bufs[1].ptr + 4, bufs[2].ptr]
            else:
                ptrs = [b.ptr for b in bufs]
            cx = (C.c_void_p * 3)(*[e.ctx for e in engs])
            bp = (C.c_void_p * 3)(*ptrs)
            lib.wga_emu_peer_copies_reset()
            assert lib.wga_reduce_scatter_i32(cx, 3, bp, count) == 0
            for e in engs:
                e.sync()
            if staged and count >= 7:
                assert [lib.wga_emu_peer_copies_in_flight(g) for g in range(3)] == [2, 2, 2]
            elif not staged:
                assert [lib.wga_emu_peer_copies_in_flight(g) for g in range(3)] == [0, 0, 0]     # nothing is copied
            total = host[0][:count].astype(np.int64) + host[1][:count] + host[2][:count]
            for g in range(3):
                lo, hi = count * g // 3, count * (g + 1) // 3
                got = bufs[g].numpy()
                got = got[1:count + 1] if (count == 1000 and g == 1) else got[:count]
                assert (got[lo:hi] == total[lo:hi]).all(), (staged, count, g)
                mask = np.ones(count, dtype=bool)
                mask[lo:hi] = False
                assert (got[mask] == host[g][:count][mask]).all(), (staged, count, g)
    one = (C.c_void_p * 1)(engs[0].ctx)
    b1 = (C.c_void_p * 1)(bufs[0].ptr)
    assert lib.wga_reduce_scatter_i32(one, 1, b1, 5) == 0
    dup = (C.c_void_p * 2)(engs[0].ctx, engs[0].ctx)
    b2 = (C.c_void_p * 2)(bufs[0].ptr, bufs[1].ptr)
    assert lib.wga_reduce_scatter_i32(dup, 2, b2, 5) == -1
    for e in engs:
        e.close()


def test_bgzf_inflate(emu):
    pc.check_bgzf_inflate(emu)


def test_bgzf_deflate(emu):
    pc.check_bgzf_deflate(emu)


def test_bgzf_deflate_random_inputs(emu):
    """K18 on whatever bytes: alphabets of 1 to 256 symbols with flat, skewed and geometric frequencies (a geometric one is what
    drives a Huffman tree past 15 levels), lengths around the member and chunk borders, every alignment — the stream is valid BGZF
    and inflates to the input"""
    from hypothesis import given, settings, strategies as st
    sizes = st.one_of(st.integers(0, 600), st.sampled_from([32767, 32768, 32769, 65535, 65536, 65537, 128 * 255, 128 * 255 + 3]),
                      st.integers(0, 140000))

    @settings(max_examples=40, deadline=None)
    @given(st.integers(0, 2 ** 31), sizes, st.integers(1, 256), st.sampled_from(["flat", "skewed", "geometric"]), st.integers(0, 3), st.integers(0, 3),
           st.booleans())
    def run(seed, n, k, shape, ia, oa, marker):
        rng = np.random.default_rng(seed)
        alphabet = rng.permutation(256)[:k]
        if shape == "flat":
            p = np.ones(k)
        elif shape == "skewed":
            p = rng.random(k) ** 6 + 1e-9
        else:
            p = 0.62 ** np.arange(k) + 1e-12
        data = alphabet[rng.choice(k, n, p=p / p.sum())].astype(np.uint8).tobytes()
        d_in = emu.upload(np.frombuffer(b"\x55" * ia + data + b"\x66" * 8, dtype=np.uint8))
        cap = int(emu.lib.wga_bgzf_bound(n))
        out = emu.empty(oa + cap + 8, np.uint8).fill(0x23)
        _, used = emu.bgzf_compress(d_in, n, out=out, eof_marker=marker, in_offset=ia, out_offset=oa)
        got = out.numpy()
        assert used <= cap and (got[:oa] == 0x23).all() and (got[oa + used:] == 0x23).all()
        assert pc.bgzf_check_stream(got[oa:oa + used].tobytes(), data, marker) == (n + 32767) // 32768

    run()
