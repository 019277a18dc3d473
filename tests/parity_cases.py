"""Parity checks of the HIP path against the CPU oracle, written once and run twice:
   * tests/test_emu_parity.py — same kernel source on the SIMT emulator (`-m "not gpu"`)
   * tests/test_gpu_parity.py — libwgahip.so on a real MI355X (`-m gpu`), through the C-ABI.
Bit-exact: every comparison is equality of integers / bytes.
"""
import os

import numpy as np

import oracle_py as orc
from wgatools_amd import engine, synth

NONE = int(engine.NONE)


def rec_ops(b, i):
    return b["ops"][int(b["op_off"][i]):int(b["op_off"][i + 1])]


def rec_text(b, i):
    return "cg:Z:" + synth.cigar_text(rec_ops(b, i))


# ------------------------------------------------------------------------------------------------
# K1
# ------------------------------------------------------------------------------------------------
def check_stat(eng, b, sample=None):
    """wga_cigar_stat == parse_paf_to_cigar (cigar.rs:629-707) for every (sampled) record"""
    batch = eng.make_batch(b["ops"], b["op_off"], b["strand_neg"])
    counts, diag, _ = eng.cigar_stat(batch)
    c, d = counts.numpy(), diag.numpy()
    # stat totals = column sums of the counters
    tot = eng.counts_total(batch.n, counts).numpy()
    cm = c.view(np.uint64).reshape(batch.n, 11) if batch.n else np.zeros((0, 11), dtype=np.uint64)
    assert (tot == cm.sum(axis=0, dtype=np.uint64)).all(), (tot, cm.sum(axis=0))
    idx = range(batch.n) if sample is None else sample
    for i in idx:
        try:
            exp = orc.parse_paf_to_cigar(rec_text(b, i), b["strand_neg"][i])
            assert d["bad_op_idx"][i] == engine.NONE, (i, d[i])
            assert tuple(int(x) for x in c[i]) == exp, (i, exp, c[i])
        except orc.OracleError as e:
            # the oracle rejected an op: the kernel must point at that very op
            assert e.kind == 2
            k = int(d["bad_op_idx"][i])
            assert k != NONE, (i, e.message)
            w = int(rec_ops(b, i)[k])
            ch = synth.OP_CHARS[w & 15] if (w & 15) < 9 else "B"
            assert e.arg == ch, (i, k, e.arg, ch)
            # and no earlier op is rejected
            for w2 in rec_ops(b, i)[:k].tolist():
                assert (w2 & 15) in (0, 1, 2, 7, 8, 9, 10)
    return counts, diag


# ------------------------------------------------------------------------------------------------
# K2
# ------------------------------------------------------------------------------------------------
DEFAULT_EXPAND_VARIANT = {"0": 0, "3": 3}.get(os.environ.get("WGA_EXPAND_VARIANT", ""), -1)   # -1: the library picks by the batch


def run_paf2maf(eng, b, pre=None, force_slow=0, fill=0x23, no_table=0, variant=None):
    """stat -> layout -> expand; returns host copies.  variant: 0 = v1 of the row kernel, 3 = the streaming kernel; None =
    whatever the context runs by default"""
    n = len(b["strand_neg"])
    batch = eng.make_batch(b["ops"], b["op_off"], b["strand_neg"])
    eng.set_param("expand_force_slow", force_slow)
    eng.set_param("expand_no_table", no_table)
    if variant is not None:
        eng.set_param("expand_variant", variant)
    counts, diag, tws = eng.cigar_stat(batch)
    tl, ql = eng.upload(b["t_src_len"]), eng.upload(b["q_src_len"])
    to, qo = eng.upload(b["t_src_off"]), eng.upload(b["q_src_off"])
    if pre is not None:
        pt, pq, po = (eng.upload(np.asarray(x, dtype=np.uint32)) for x in pre)
    else:
        pt = pq = po = None
    tro, qro, reco = eng.paf2maf_layout(n, counts, tl, ql, pt, pq, po)
    total = int(reco.numpy()[-1])
    out = eng.empty(total + 64, np.uint8).fill(fill)
    tp, qp = eng.upload(b["t_pool"]), eng.upload(b["q_pool"])
    eng.paf2maf_expand(batch, counts, tws, tp, len(b["t_pool"]), to, tl, qp, len(b["q_pool"]), qo,
                       ql, out, tro, qro, diag)
    eng.set_param("expand_force_slow", 0)
    eng.set_param("expand_no_table", 0)
    if variant is not None:
        eng.set_param("expand_variant", DEFAULT_EXPAND_VARIANT)
    return dict(out=out.numpy(), t_row_off=tro.numpy(), q_row_off=qro.numpy(),
                rec_off=reco.numpy(), counts=counts.numpy(), diag=diag.numpy(), total=total)


def check_drain_min_settings(eng, b):
    """RowSrc::drain_min only decides WHEN a wave emits its queued gap-touching chunks: every setting, and the
    library's own choice, must give the bytes the oracle gives"""
    ref = None
    for dm in (0, 1, 16, 32, 64):
        eng.set_param("expand_drain_min", dm)
        try:
            check_paf2maf(eng, b, variant=0)
            r = run_paf2maf(eng, b, variant=0)
        finally:
            eng.set_param("expand_drain_min", 0)
        assert eng.get_param("expand_drain_min") == (dm if dm else eng.get_param("expand_drain_min"))
        if ref is None:
            ref = r["out"].copy()
        assert (r["out"] == ref).all(), dm
    for bad in (-1, 65):
        try:
            eng.set_param("expand_drain_min", bad)
        except Exception:
            pass
        else:
            raise AssertionError("expand_drain_min %d accepted" % bad)
    assert eng.get_param("expand_variant") == DEFAULT_EXPAND_VARIANT


def oracle_rows(b, i):
    """what converter.rs:219-235 does for one record, through the oracle"""
    t = b["t_pool"][int(b["t_src_off"][i]):int(b["t_src_off"][i] + b["t_src_len"][i])].tobytes()
    q = b["q_pool"][int(b["q_src_off"][i]):int(b["q_src_off"][i] + b["q_src_len"][i])].tobytes()
    if b["strand_neg"][i]:
        q = orc.reverse_complement(q)  # may raise InvalidBase
    return orc.parse_cigar_to_insert(rec_text(b, i), t, q)  # may raise CigarOpInvalid / panic


def check_paf2maf(eng, b, pre=None, force_slow=0, sample=None, no_table=0, variant=None):
    r = run_paf2maf(eng, b, pre=pre, force_slow=force_slow, no_table=no_table, variant=variant)
    out, n = r["out"], len(b["strand_neg"])
    covered = np.zeros(len(out), dtype=bool)
    idx = range(n) if sample is None else sample
    for i in idx:
        d = r["diag"][i]
        try:
            et, eq = oracle_rows(b, i)
        except orc.OracleError as e:
            if e.kind == 4:  # InvalidBase: first offender in reversed order
                pos = int(d["bad_base_pos"])
                assert pos != NONE, (i, e.message)
                raw = b["q_pool"][int(b["q_src_off"][i] + b["q_src_len"][i]) - 1 - pos]
                assert chr(raw) == e.arg, (i, pos, chr(raw), e.arg)
            elif e.kind == 2:
                assert int(d["bad_op_idx"]) != NONE, (i, e.message)
            else:
                assert e.kind == 6 and int(d["panic_op_idx"]) != NONE, (i, e.message, d)
            continue
        assert int(d["bad_op_idx"]) == NONE and int(d["panic_op_idx"]) == NONE and \
            int(d["bad_base_pos"]) == NONE, (i, d)
        to, qo = int(r["t_row_off"][i]), int(r["q_row_off"][i])
        gt, gq = out[to:to + len(et)].tobytes(), out[qo:qo + len(eq)].tobytes()
        if gt != et or gq != eq:
            k = next((j for j in range(len(et)) if gt[j] != et[j]), None)
            k2 = next((j for j in range(len(eq)) if gq[j] != eq[j]), None)
            raise AssertionError("record %d rows differ: t@%s q@%s (neg=%d, len=%d)" % (
                i, k, k2, b["strand_neg"][i], len(et)))
        covered[to:to + len(et)] = True
        covered[qo:qo + len(eq)] = True
        # geometry: rows sit where the layout says (String::insert_str lengths)
        assert len(et) == int(b["t_src_len"][i]) + int(r["counts"][i]["ins_bp"] + r["counts"][i]["inv_ins_bp"])
        assert len(eq) == int(b["q_src_len"][i]) + int(r["counts"][i]["del_bp"] + r["counts"][i]["inv_del_bp"])
    if sample is None:
        # nothing outside the rows of clean records was touched (no stray / RMW stores)
        clean = np.ones(n, dtype=bool)
        for f in ("bad_op_idx", "panic_op_idx", "bad_base_pos"):
            clean &= r["diag"][f] == engine.NONE
        if clean.all():
            assert (out[~covered] == 0x23).all()
    return r


# ------------------------------------------------------------------------------------------------
# hand-made batches for the edge cases
# ------------------------------------------------------------------------------------------------
def batch_from_texts(eng, cigars, strands, t_seqs, q_seqs, pad=0):
    """records given as text; slices are laid back to back (optionally `pad` bytes apart) in the
    pools so that the first / last records touch the pool edges"""
    from helpers import pack_records
    ops, off, errs = pack_records(eng, cigars)
    assert all(e == 0 for e in errs), errs
    def pool(seqs):
        offs, buf, p = [], bytearray(), 0
        for s in seqs:
            offs.append(p)
            buf += s + b"N" * pad
            p += len(s) + pad
        if pad and buf:
            del buf[-pad:]
        return np.frombuffer(bytes(buf) or b"N", dtype=np.uint8).copy(), np.array(offs, dtype=np.uint64)
    tp, to = pool(t_seqs)
    qp, qo = pool(q_seqs)
    return dict(ops=ops, op_off=off, strand_neg=np.array(strands, dtype=np.uint8), t_pool=tp,
                q_pool=qp, t_src_off=to, q_src_off=qo,
                t_src_len=np.array([len(s) for s in t_seqs], dtype=np.uint64),
                q_src_len=np.array([len(s) for s in q_seqs], dtype=np.uint64))


def rand_seq(rng, n, alphabet=b"ACGTacgtNn"):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return a[rng.integers(0, len(a), size=n)].tobytes()


def consumption(cigar):
    """(target bases, query bases) a CIGAR text consumes (M = X / D / I)"""
    import re
    t = q = 0
    for n, op in re.findall(r"(\d+)(\D)", cigar):
        n = int(n)
        if op in "M=X":
            t += n
            q += n
        elif op == "D":
            t += n
        elif op == "I":
            q += n
    return t, q


def edge_case_batch(eng, seed=7):
    """zero-length ops, 1-op records, leading/trailing indels, slices longer / shorter than the
    CIGAR consumes, both strands, records touching both pool edges"""
    rng = np.random.default_rng(seed)
    cigars = [
        "5=", "1X", "1I", "1D", "0M3=0I2X0D", "3I5=2D", "2D5=3I", "7=1X7=1X7=1X7=", "40=",
        "16=", "15=1I", "1D15=", "17=3I17=3D17=", "1=1I1=1D1=1I1=1D1=1I1=1D", "100I", "100D",
        "33=17I5X19D64=", "8M8I8M8D8M",
    ]
    strands = [i % 2 for i in range(len(cigars))]
    t_seqs, q_seqs = [], []
    for c in cigars:
        t, q = consumption(c)
        t_seqs.append(rand_seq(rng, t))
        q_seqs.append(rand_seq(rng, q))
    # slices longer than the CIGAR consumes (tails are appended) and shorter (rows end early)
    extra = [("10=2I10=", 0, 27, 30), ("10=2I10=", 1, 20, 40), ("12=", 1, 5, 12), ("12=", 0, 12, 3),
             ("4=2D4=", 1, 10, 6), ("30=", 0, 64, 64), ("30=", 1, 64, 64)]
    for c, s, tl, ql in extra:
        cigars.append(c)
        strands.append(s)
        t_seqs.append(rand_seq(rng, tl))
        q_seqs.append(rand_seq(rng, ql))
    return batch_from_texts(eng, cigars, strands, t_seqs, q_seqs)


def consistent_batch(seed, n, mean_ops):
    """records whose sequences AGREE with their CIGAR (= columns equal, X columns differ, upper-case ACGT), ops drawn
    as  = run, edit, = run, edit ... with no two neighbouring ops of one kind: the MAF rows paf2maf writes for them
    read back (parse_maf_seq_to_cigar, cigar.rs:344-432) as exactly the CIGAR they came from"""
    rng = np.random.default_rng(seed)
    nops = np.maximum(1, rng.poisson(mean_ops, n)) | 1            # odd: starts and ends with an = run
    tot = int(nops.sum())
    op_off = np.zeros(n + 1, dtype=np.uint64)
    op_off[1:] = np.cumsum(nops)
    pos = np.arange(tot) - np.repeat(op_off[:-1].astype(np.int64), nops)
    is_eq = (pos & 1) == 0
    kind = rng.choice(np.array([8, 1, 2], dtype=np.uint32), tot, p=[0.6, 0.2, 0.2])
    code = np.where(is_eq, np.uint32(7), kind).astype(np.uint32)
    ln = np.where(is_eq, rng.geometric(1 / 24.0, tot), np.where(code == 8, rng.integers(1, 3, tot), rng.geometric(1 / 3.0, tot)))
    ln = np.where((code != 8) & ~is_eq & (rng.random(tot) < 0.01), rng.integers(50, 2000, tot), ln).astype(np.uint32)
    ops = (ln << np.uint32(4)) | code
    # per-column classes of the whole batch
    ccls = np.repeat(code, ln)
    ncol = len(ccls)
    tb = rng.integers(0, 4, ncol).astype(np.uint8)
    qb = np.where(ccls == 7, tb, np.where(ccls == 8, (tb + rng.integers(1, 4, ncol)) & 3, rng.integers(0, 4, ncol))).astype(np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    has_t, has_q = ccls != 1, ccls != 2
    col_rec = np.repeat(np.arange(n), np.add.reduceat(ln.astype(np.int64), op_off[:-1].astype(np.int64)))
    strand = rng.integers(0, 2, n).astype(np.uint8)
    t_len = np.bincount(col_rec[has_t], minlength=n).astype(np.uint64)
    q_len = np.bincount(col_rec[has_q], minlength=n).astype(np.uint64)
    t_flat, q_flat = lut[tb[has_t]], lut[qb[has_q]]
    pad = np.frombuffer(b"N" * 32, dtype=np.uint8)
    t_off = np.zeros(n, dtype=np.uint64)
    t_off[1:] = np.cumsum(t_len)[:-1]
    q_off = np.zeros(n, dtype=np.uint64)
    q_off[1:] = np.cumsum(q_len)[:-1]
    # '-' strand: the pool holds the reverse complement of what the row shows
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    q_pool_core = q_flat.copy()
    for i in np.flatnonzero(strand):
        a, z = int(q_off[i]), int(q_off[i] + q_len[i])
        q_pool_core[a:z] = comp[q_flat[a:z]][::-1]
    return dict(ops=ops, op_off=op_off, strand_neg=strand, t_pool=np.concatenate([pad, t_flat, pad]),
                q_pool=np.concatenate([pad, q_pool_core, pad]), t_src_off=t_off + np.uint64(32), t_src_len=t_len,
                q_src_off=q_off + np.uint64(32), q_src_len=q_len)


def check_paf2maf_maf2paf_roundtrip(eng, seed, n, mean_ops):
    """paf2maf (K1 + layout + K2) then maf2paf's walk (K3 runs -> K11 packed ops) over the rows it wrote gives back
    the op stream and the counters it started from: two kernels that share no code agree"""
    b = consistent_batch(seed, n, mean_ops)
    r = run_paf2maf(eng, b)
    assert (r["diag"]["bad_op_idx"] == NONE).all() and (r["diag"]["panic_op_idx"] == NONE).all()
    c = r["counts"]
    cols = (c["match"] + c["mismatch"] + c["ins_bp"] + c["inv_ins_bp"] + c["del_bp"] + c["inv_del_bp"]).astype(np.uint64)
    rows = eng.upload(r["out"])
    d_t, d_q, d_c = eng.upload(r["t_row_off"]), eng.upload(r["q_row_off"]), eng.upload(cols)
    d_s = eng.upload(b["strand_neg"])
    counts2, run_cnt = eng.maf_pair_stat(n, rows, d_t, d_q, d_c, d_s)
    run_off = eng.exclusive_scan_u64(n, run_cnt)
    ne = int(run_off.numpy()[-1])
    runs = eng.empty(ne + 1, np.uint64).fill(0)
    counts2, _ = eng.maf_pair_stat(n, rows, d_t, d_q, d_c, d_s, counts=counts2, run_cnt=run_cnt, runs=runs, run_off=run_off)
    ocnt = eng.maf_runs_ops(n, ne, runs, run_off, d_c)
    ooff = eng.exclusive_scan_u64(n, ocnt)
    oo = ooff.numpy()
    ops2 = eng.empty(int(oo[-1]) + 4, np.uint32).fill(0)
    eng.maf_runs_ops(n, ne, runs, run_off, d_c, out=ops2, out_off=ooff)
    assert (oo == b["op_off"]).all(), "op counts per record differ"
    got = ops2.numpy()[: int(oo[-1])]
    if not (got == b["ops"]).all():
        k = int(np.flatnonzero(got != b["ops"])[0])
        raise AssertionError("op %d differs: %#x vs %#x" % (k, int(got[k]), int(b["ops"][k])))
    c2 = counts2.numpy()
    for f in c.dtype.names:
        assert (c2[f] == c[f]).all(), f
    return int(oo[-1])


def wide_tile_batch(eng, seed=3):
    """records whose tiles hold 65 536 .. 2^31 columns (one long D / I op each, next to ordinary ops): the u32
    instance of the row kernels; the short records around them stay in narrow tiles"""
    rng = np.random.default_rng(seed)
    cigars = ["20=", "30=70000D25=3I8=", "9=2X9=", "12=100000I12=", "5=1I5=", "40=66000D3X90000I17="]
    strands = [0, 1, 1, 0, 0, 1]
    t_seqs, q_seqs = [], []
    for c in cigars:
        t, q = consumption(c)
        t_seqs.append(rand_seq(rng, t))
        q_seqs.append(rand_seq(rng, q))
    return batch_from_texts(eng, cigars, strands, t_seqs, q_seqs)


def window_kernel_cases(eng, variant=3):
    """a row kernel (expand_variant 3 = the streaming kernel, 0 = v1) against the oracle: random mixtures, edge cases with odd row alignments,
    records over many tiles, many records per tile (more than 16 and more than 64 record segments in a tile), wide tiles,
    the op-serial walk behind it"""
    from wgatools_amd import synth
    for seed, n, mean, pool, use_m in [(1, 12, 700, 50000, False), (2, 40, 60, 20000, False), (3, 300, 3, 5000, True),
                                       (4, 3, 5000, 200000, True)]:
        b = synth.make_paf_batch(seed, n, mean, pool, use_m=use_m)
        rng = np.random.default_rng(seed)
        check_paf2maf(eng, b, variant=variant)
        check_paf2maf(eng, b, pre=(rng.integers(0, 40, n), rng.integers(0, 40, n), rng.integers(0, 5, n)), variant=variant)
    e = edge_case_batch(eng)
    ne = len(e["strand_neg"])
    rng = np.random.default_rng(5)
    check_paf2maf(eng, e, variant=variant)
    check_paf2maf(eng, e, pre=(rng.integers(0, 33, ne), rng.integers(0, 33, ne), rng.integers(0, 3, ne)), variant=variant)
    check_paf2maf(eng, e, force_slow=1, variant=variant)
    check_paf2maf(eng, wide_tile_batch(eng), variant=variant)
    for seed in (20, 21):
        check_paf2maf(eng, synth.make_paf_batch(seed, 700, 2, 8000), variant=variant)      # hundreds of records per tile
    for seed in (30, 31, 32, 33):
        n = int(np.random.default_rng(seed).integers(1, 50))
        mean = int(np.random.default_rng(seed + 1).integers(1, 2000))
        b = synth.make_paf_batch(seed, n, mean, 100000, use_m=bool(seed & 1))
        rng = np.random.default_rng(seed)
        check_paf2maf(eng, b, pre=(rng.integers(0, 130, n), rng.integers(0, 130, n), rng.integers(0, 5, n)), variant=variant)


def stream_kernel_cases(eng):
    """the streaming row kernel (expand_variant 3) beyond window_kernel_cases: every job length, records that are not clean
    (left to v1) between clean ones inside one job, long records over many jobs, dense gap events (FIFO pressure, many
    events per granule), invalid bases on '-' strand rows, slices at both pool edges"""
    from wgatools_amd import synth
    try:
        for jt in (1, 2, 4, 32):
            eng.set_param("expand_job_tiles", jt)
            b = synth.make_paf_batch(40 + jt, 9, 3000, 300_000)
            rng = np.random.default_rng(jt)
            check_paf2maf(eng, b, pre=(rng.integers(0, 130, 9), rng.integers(0, 130, 9), rng.integers(0, 5, 9)), variant=3)
            assert eng.get_param("expand_variant_used") == 3 and eng.get_param("expand_stream_left_to_v1") == 0
            # records 2 and 5 get a tail (slice longer than the CIGAR consumes), record 6 a short slice: their tiles are v1's
            u = dict(b)
            tl, ql = b["t_src_len"].copy(), b["q_src_len"].copy()
            tl[2] += 7
            ql[5] += 3
            tl[6] -= min(5, int(tl[6]))
            u["t_src_len"], u["q_src_len"] = tl, ql
            check_paf2maf(eng, u, variant=3)
            assert eng.get_param("expand_stream_left_to_v1") > 0
            check_paf2maf(eng, dense_indel_batch(eng), variant=3)
        eng.set_param("expand_job_tiles", 4)
        big = synth.make_paf_batch(12, 1, 40_000, 400_000, sigma=0.01)  # one record over ~40 tiles = 10 jobs
        check_paf2maf(eng, big, variant=3)
        bad = synth.make_paf_batch(13, 3, 3000, 100_000)                # invalid bases on '-' strand rows
        bad["strand_neg"][:] = 1
        qp = bad["q_pool"].copy()
        for r_, f in ((0, 3), (1, 2), (1, 5)):
            qp[int(bad["q_src_off"][r_] + bad["q_src_len"][r_] * f // 7)] = ord("R-x"[f % 3])
        bad["q_pool"] = qp
        r = check_paf2maf(eng, bad, variant=3)
        assert int(r["diag"]["bad_base_pos"][0]) != int(r["diag"]["bad_base_pos"][2])
        # slices that begin at byte 0 and end at the last byte of their pools (no padding around them)
        cig = ["700=3I900=2D650=", "1500=1X200=4D300="]
        tq = [consumption(c) for c in cig]
        rs = np.random.default_rng(3)
        for strands in ([0, 0], [1, 1]):
            check_paf2maf(eng, batch_from_texts(eng, cig, strands, [rand_seq(rs, t) for t, _ in tq], [rand_seq(rs, q) for _, q in tq], pad=0),
                          variant=3)
    finally:
        eng.set_param("expand_job_tiles", 4)


def dense_indel_batch(eng, seed=9):
    """stretches with an indel every few columns: more than 63 gap ops inside one 4 KB output window, three and more gap ops
    inside sixteen columns, zero-length gap ops, gaps of thousands of columns (whole-dash windows)"""
    rng = np.random.default_rng(seed)
    def dense(n):
        out = []
        for _ in range(n):
            out.append("%d=" % rng.integers(1, 4))
            out.append("%d%s" % (rng.integers(0, 3), "ID"[int(rng.integers(0, 2))]))
        return "".join(out) + "7="
    cigars = [dense(900), "5=" + dense(300) + "9000D3=" + dense(100) + "12000I4=", dense(2500), "3=1I1D1I1D2=0I0D5=" * 40 + "1="]
    strands = [0, 1, 1, 0]
    t_seqs, q_seqs = [], []
    for c in cigars:
        t, q = consumption(c)
        t_seqs.append(rand_seq(rng, t))
        q_seqs.append(rand_seq(rng, q))
    return batch_from_texts(eng, cigars, strands, t_seqs, q_seqs, pad=40)


# ------------------------------------------------------------------------------------------------
# K5 pafcov
# ------------------------------------------------------------------------------------------------
def sprinkle_ops(rng, b, frac=0.02, codes=(3, 4, 5, 6, 11)):
    """replace a few ops by N/S/H/P/other (legal for pafcov / pafpseudo)"""
    ops = b["ops"].copy()
    k = max(1, int(len(ops) * frac))
    idx = rng.integers(0, len(ops), k)
    ops[idx] = (ops[idx] & ~np.uint32(15)) | rng.choice(np.array(codes, dtype=np.uint32), k)
    nb = dict(b)
    nb["ops"] = ops
    return nb


def text_any(ops_slice):
    chars = "MIDNSHP=XIDB"
    return "cg:Z:" + "".join("%d%s" % (w >> 4, chars[w & 15]) for w in ops_slice.tolist())


def check_pafcov_long_ops(eng):
    """K5: a window (16 384 counters since round 6, 8 192 before: both sizes' borders are hit) replays only the lanes of a record
    segment whose ops can mark inside it.  Ops far longer than a window (a segment across 60+ windows, windows without any mark
    inside), ops ending exactly on window borders, zero-length ops, a record that runs past its target's end, two targets back
    to back in the coverage array"""
    mk = lambda p: [(int(ln) << 4) | int(c) for c, ln in p]
    for W in (8192, 16384):
        _check_pafcov_window_borders(eng, W, mk)
    # a tile that advances 2^31 bases and more (N ops of the longest packed length): the list pass measures its segments one by
    # one in 64-bit positions, the replay walks such a piece in 64-bit positions too; the record goes on into the next tile
    # (whose look-back brings a sum beyond 2^31) and far beyond its target's end; ordinary records in front and behind
    big = (1 << 28) - 1
    recs = [mk([(7, 40), (8, 1)] * 300),
            mk([(7, 100)] + [(3, big)] * 9 + [(7, 50)] + [(7, 3), (2, 1)] * 600),
            mk([(7, 25), (1, 2), (0, 9000)] * 40),
            mk([(7, 5)] + [(3, big)] * 3 + [(7, 7), (2, big), (7, 1)])]       # beyond 2^30 but below 2^31: a wide piece of a narrow tile
    ops = np.array([o for r in recs for o in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    b = dict(ops=ops, op_off=off, strand_neg=np.zeros(len(recs), dtype=np.uint8))
    check_pafcov(eng, b, [0, 1, 0, 1], [30, 10, 2000, 70000], [400000, 90000])
    check_pafcov(eng, b, [0, 1, 0, 1], [30, 10, 2000, 70000], [400000, 90000], split=True)
    # WIDE pieces (a segment that advances 2^27 bases or more: the replay reads the piece itself through its descriptor) in a
    # tile of more pieces than its eight slots hold, so that some of them stand in the list regions: forty records of three ops in
    # one tile, each with an N op of 2^28 - 1 or 2^27 bases behind five counted ones, over targets of a few windows; around them
    # ordinary records whose pieces are narrow
    recs = [mk([(7, 9), (8, 1)] * 20)]
    recs += [mk([(7, 5 + k), (3, big if k % 3 else 1 << 27), (7, 5)]) for k in range(40)]
    recs += [mk([(7, 30), (2, 4), (0, 17)] * 50)]
    ops = np.array([o for r in recs for o in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    b = dict(ops=ops, op_off=off, strand_neg=np.zeros(len(recs), dtype=np.uint8))
    tid = [0] + [k % 2 for k in range(40)] + [1]
    ts = [77] + [(k * 2711) % 60000 for k in range(40)] + [123]
    check_pafcov(eng, b, tid, ts, [70000, 100000])
    check_pafcov(eng, b, tid, ts, [70000, 100000], split=True)


def _check_pafcov_window_borders(eng, W, mk):
    recs = [mk([(7, 300000), (2, 5), (0, 200001), (8, 1), (1, 7), (7, 9)]),
            mk([(7, W), (7, W), (2, W), (7, 1), (8, W - 1), (7, 0), (0, 3 * W)]),
            mk([(7, 3), (1, 2)] * 700 + [(7, 100000)] + [(8, 1), (7, 2)] * 300),
            mk([(7, 5000), (3, 40000), (7, 5000), (2, 70000), (7, 20000)]),      # N and D move without counting
            mk([(7, 60000)] * 3)]                                                   # runs past the end of its target
    ops = np.array([o for r in recs for o in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    n = len(recs)
    b = dict(ops=ops, op_off=off, strand_neg=np.zeros(n, dtype=np.uint8))
    check_pafcov(eng, b, [0, 0, 1, 1, 1], [100, W * 3 - 1, 0, 50000, 150000], [600000, 250000])
    check_pafcov(eng, b, [0, 0, 1, 1, 1], [100, W * 3 - 1, 0, 50000, 150000], [600000, 250000], split=True)


def check_pafcov_look_back(eng):
    rng = np.random.default_rng(77)
    lens = [2048, 1, 4096 + 5, 70 * 2048 + 17, 3, 2043, 2048 * 3, 2, 5000, 2048 - 5, 1024, 1024]     # ops per record (K5's tile: 2048)
    recs = []
    for n_ops in lens:
        code = rng.choice(np.array([7, 7, 7, 8, 1, 2, 0, 3, 4], dtype=np.uint32), n_ops)
        ln = rng.integers(0, 9, n_ops).astype(np.uint32)
        recs.append((ln << 4) | code)
    ops = np.concatenate(recs)
    off = np.cumsum([0] + lens).astype(np.uint64)
    n = len(lens)
    b = dict(ops=ops, op_off=off, strand_neg=np.zeros(n, dtype=np.uint8))
    tid = [0, 1, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1]
    starts = [0, 5, 100, 40, 9000, 20000, 250000, 7, 30000, 100, 41000, 290000]
    check_pafcov(eng, b, tid, starts, [60000, 600000])
    check_pafcov(eng, b, tid, starts, [60000, 600000], split=True)


def check_pafcov_random(eng, seed, cases):
    """K5 on random shapes: records of 1 .. 5 000 ops with the sizes around the tile (1 024 ops), lane (16) and step (256)
    borders, ops of every class, a few of them far longer than a window, 1 .. 4 targets short enough to clip records,
    records that start beyond their target, one call or two"""
    rng = np.random.default_rng(seed)
    sizes = [1, 1, 1, 2, 3, 15, 16, 17, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 5000]
    codes = np.array([7, 7, 7, 0, 8, 1, 2, 3, 4, 5, 6, 11], dtype=np.uint32)
    for _ in range(cases):
        n = int(rng.integers(1, 40))
        lens = [int(rng.choice(sizes)) for _ in range(n)]
        recs = []
        for m in lens:
            c = rng.choice(codes, m)
            ln = np.where(rng.random(m) < 0.02, rng.integers(0, 120000, m), rng.integers(0, 40, m)).astype(np.uint32)
            recs.append((ln << 4) | c)
        b = dict(ops=np.concatenate(recs).astype(np.uint32), op_off=np.cumsum([0] + lens).astype(np.uint64),
                 strand_neg=np.zeros(n, dtype=np.uint8))
        nt = int(rng.integers(1, 5))
        tlen = [int(rng.integers(1, 400000)) for _ in range(nt)]
        tid = rng.integers(0, nt, n)
        ts = [int(rng.integers(0, int(tlen[t] * 1.1) + 1)) for t in tid]
        check_pafcov(eng, b, list(tid), ts, tlen, align=int(rng.choice([1, 4])), split=bool(rng.integers(0, 2)))


def check_pafcov_many_small_targets(eng, seed=5, nt=400):
    """hundreds of targets shorter than a window (8 192 counters), a few of no bases at all, packed without alignment: a
    window of the counting replay holds dozens of range borders (its counter-by-counter walk, every range a chain of its
    own), and records that start in one target's counters run over its end"""
    rng = np.random.default_rng(seed)
    tlen = [int(x) for x in rng.choice([0, 1, 2, 7, 30, 100, 300, 1000, 9000], nt)]
    n = 3 * nt
    tid = [int(x) for x in rng.integers(0, nt, n)]
    codes = np.array([7, 7, 0, 8, 1, 2, 3], dtype=np.uint32)
    recs, lens = [], []
    for _ in range(n):
        m = int(rng.integers(1, 12))
        recs.append((rng.integers(0, 60, m).astype(np.uint32) << 4) | rng.choice(codes, m))
        lens.append(m)
    b = dict(ops=np.concatenate(recs).astype(np.uint32), op_off=np.cumsum([0] + lens).astype(np.uint64),
             strand_neg=np.zeros(n, dtype=np.uint8))
    ts = [int(rng.integers(0, tlen[t] + 3)) for t in tid]
    check_pafcov(eng, b, tid, ts, tlen, align=1)
    check_pafcov(eng, b, tid, ts, tlen, align=1, split=True)


def check_pafcov(eng, b, target_id, t_start, target_len, align=4, split=False, shuffle=None):
    """both protocols against update_cov_vec: accumulate() per batch + finalize(), and the last batch through
    accumulate_final() (marks and the marks -> counts scan in one pass).  The targets' ranges lie in the array in the order
    `shuffle` gives (default: by index, or a seeded permutation for three and more targets), `align`-aligned with the gaps and
    the array's tail holding a pattern that must still be there afterwards"""
    n = len(b["strand_neg"])
    nt = len(target_len)
    order = list(range(nt))
    if shuffle is None and nt >= 3:
        np.random.default_rng(nt * 7919 + int(sum(target_len)) % 1000).shuffle(order)
    elif shuffle is not None:
        order = list(shuffle)
    cov_off = np.zeros(nt, dtype=np.uint64)
    p = 0
    for t in order:
        p = (p + align - 1) // align * align
        cov_off[t] = p
        p += int(target_len[t])
    total = p + 8
    PAT = np.int32(0x5A5A5A5A)
    init = np.full(total, PAT, dtype=np.int32)
    for t in range(nt):
        init[int(cov_off[t]):int(cov_off[t]) + int(target_len[t])] = 0
    d_off, d_len = eng.upload(cov_off), eng.upload(np.asarray(target_len, dtype=np.uint64))
    cuts = [0, n // 3, n] if split and n >= 3 else [0, n]      # several accumulate() calls, one finalize()
    exp = [np.zeros(int(l), dtype=np.uint64) for l in target_len]
    for i in range(n):
        orc.update_cov_vec(exp[target_id[i]], text_any(rec_ops(b, i)), int(t_start[i]))
    got = None
    for fused in (False, True):
        cov = eng.upload(init)
        for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
            a, z = int(b["op_off"][lo]), int(b["op_off"][hi])
            batch = eng.make_batch(b["ops"][a:z], b["op_off"][lo:hi + 1] - np.uint64(a), b["strand_neg"][lo:hi])
            d_tid = eng.upload(np.asarray(target_id[lo:hi], dtype=np.uint32))
            d_ts = eng.upload(np.asarray(t_start[lo:hi], dtype=np.uint64))
            if fused and k == len(cuts) - 2:
                eng.pafcov_accumulate_final(batch, d_tid, d_ts, d_off, d_len, nt, cov, p)
            else:
                eng.pafcov_accumulate(batch, d_tid, d_ts, d_off, d_len, cov, p)
        if not fused:
            eng.pafcov_finalize(nt, d_off, d_len, cov)
        got = cov.numpy()
        inside = np.zeros(total, dtype=bool)
        for t in range(nt):
            lo = int(cov_off[t])
            g = got[lo:lo + int(target_len[t])]
            inside[lo:lo + int(target_len[t])] = True
            assert (g.astype(np.int64) == exp[t].astype(np.int64)).all(), (
                fused, t, np.nonzero(g.astype(np.int64) != exp[t].astype(np.int64))[0][:5])
        assert (got[~inside] == PAT).all(), (fused, np.nonzero(got[~inside] != PAT)[0][:5])
    return got


# ------------------------------------------------------------------------------------------------
# K6 pafpseudo
# ------------------------------------------------------------------------------------------------
def check_pafpseudo(eng, b, base_mode, skip=None, sums_call=True, variant=None):
    """sums_call=False: the fill without wga_cigar_class_sums in front (it then computes the tile and record sums itself
    instead of taking what the class-sums call left in the context); variant: "pseudo_variant" for this call (base mode:
    3 the streaming row kernel, 0 one block per tile)"""
    if variant is not None:
        before = eng.get_param("pseudo_variant")
        eng.set_param("pseudo_variant", variant)
        try:
            return check_pafpseudo(eng, b, base_mode, skip=skip, sums_call=sums_call)
        finally:
            eng.set_param("pseudo_variant", before)
    n = len(b["strand_neg"])
    batch = eng.make_batch(b["ops"], b["op_off"], b["strand_neg"])
    sums = eng.cigar_class_sums(batch).numpy() if sums_call else {}
    # class sums against numpy
    code = b["ops"] & 15
    length = (b["ops"] >> 4).astype(np.uint64)
    cls_of = np.array([0, 1, 2, 4, 3, 4, 4, 0, 0, 1, 2, 4, 4, 4, 4, 4])
    for ci, name in enumerate(("mx", "i", "d", "s", "o")):
        v = np.where(cls_of[code] == ci, length, 0).astype(np.uint64)
        c = np.concatenate([[0], np.cumsum(v)]).astype(np.uint64)
        exp = c[b["op_off"][1:].astype(np.int64)] - c[b["op_off"][:-1].astype(np.int64)]
        if sums_call:
            assert (sums[name] == exp).all(), name
        else:
            sums[name] = exp
    skip = np.zeros(n, dtype=np.uint64) if skip is None else np.asarray(skip, dtype=np.uint64)
    # expected segments from the oracle
    exp_rows, errs = [], []
    for i in range(n):
        q = b"" if not base_mode else \
            b["q_pool"][int(b["q_src_off"][i]):int(b["q_src_off"][i] + b["q_src_len"][i])].tobytes()
        try:
            if base_mode and b["strand_neg"][i]:
                q = orc.reverse_complement(q)
            row = orc.gen_pesudo_maf_by_cigar(text_any(rec_ops(b, i)), q, bool(base_mode))
            exp_rows.append(row[int(skip[i]):])
            errs.append(None)
        except orc.OracleError as e:
            exp_rows.append(b"")
            errs.append(e)
    # host geometry (pseudomaf.rs lengths): edited length minus the trimmed head
    if base_mode:
        seg = (b["q_src_len"].astype(np.int64) - (sums["i"] + sums["s"]).astype(np.int64)
               + sums["d"].astype(np.int64))
    else:
        seg = (sums["mx"] + sums["d"]).astype(np.int64)
    seg = np.maximum(seg - skip.astype(np.int64), 0)
    rng = np.random.default_rng(n)
    gaps = rng.integers(0, 20, n)
    dst_off = np.zeros(n, dtype=np.uint64)
    p = 3
    for i in range(n):
        dst_off[i] = p
        p += int(seg[i]) + int(gaps[i])
    out = eng.empty(p + 64, np.uint8).fill(0x23)
    if base_mode:
        qp = eng.upload(b["q_pool"])
        diag = eng.pafpseudo_fill(batch, 1, qp, len(b["q_pool"]), eng.upload(b["q_src_off"]),
                                  eng.upload(b["q_src_len"]), eng.upload(skip), out,
                                  eng.upload(dst_off))
    else:
        diag = eng.pafpseudo_fill(batch, 0, None, 0, None, None, eng.upload(skip), out,
                                  eng.upload(dst_off))
    o, d = out.numpy(), diag.numpy()
    covered = np.zeros(len(o), dtype=bool)
    all_clean = True
    for i in range(n):
        if errs[i] is not None:
            all_clean = False
            if errs[i].kind == 4:
                pos = int(d["bad_base_pos"][i])
                assert pos != NONE
                raw = b["q_pool"][int(b["q_src_off"][i] + b["q_src_len"][i]) - 1 - pos]
                assert chr(raw) == errs[i].arg
            else:
                assert errs[i].kind == 6 and int(d["panic_op_idx"][i]) != NONE, (i, errs[i].message)
            continue
        assert int(d["panic_op_idx"][i]) == NONE and int(d["bad_base_pos"][i]) == NONE, (i, d[i])
        e = exp_rows[i]
        assert len(e) == int(seg[i]), (i, len(e), seg[i])
        g = o[int(dst_off[i]):int(dst_off[i]) + len(e)].tobytes()
        if g != e:
            k = next(j for j in range(len(e)) if g[j] != e[j])
            raise AssertionError("pseudo segment %d differs at %d: %r vs %r" % (
                i, k, e[max(0, k - 10):k + 10], g[max(0, k - 10):k + 10]))
        covered[int(dst_off[i]):int(dst_off[i]) + len(e)] = True
    if all_clean:
        assert (o[~covered] == 0x23).all()


def pseudo_consumption(cigar):
    """query bases pafpseudo's edit of a CIGAR consumes (M = X I S)"""
    import re
    return sum(int(n) for n, op in re.findall(r"(\d+)(\D)", cigar) if op in "M=XIS")


def pseudo_clean(rng, b, margin=64):
    """slices of exactly the length pafpseudo's edit consumes (M = X I S), `margin` bytes away from the pool's edges"""
    code, length = b["ops"] & 15, (b["ops"] >> 4).astype(np.uint64)
    v = np.where((code == 0) | (code == 7) | (code == 8) | (code == 1) | (code == 4) | (code == 9), length, 0).astype(np.uint64)
    c = np.concatenate([np.zeros(1, np.uint64), np.cumsum(v, dtype=np.uint64)])
    nb = dict(b)
    n = len(b["strand_neg"])
    nb["q_src_len"] = c[b["op_off"][1:].astype(np.int64)] - c[b["op_off"][:-1].astype(np.int64)]
    room = len(b["q_pool"]) - 2 * margin - int(nb["q_src_len"].max())
    if room <= 0:  # the pool grows
        nb["q_pool"] = np.concatenate([b["q_pool"], np.frombuffer(rand_seq(rng, 2 * margin + 1 - room), dtype=np.uint8)])
        room = 1
    nb["q_src_off"] = (margin + rng.integers(0, room, n)).astype(np.uint64)
    return nb


def pseudo_stream_cases(eng):
    """pafpseudo's rows (both modes) through the streaming row kernel ("pseudo_variant" 3): every job length; S ops; heads trimmed
    by a few columns, by whole tiles, by more than the record has; records whose slice is longer / shorter than the edit
    consumes (left to the block kernel) between clean ones; hundreds of I / S ops in a row on one column (events without
    columns: the FIFO's dedupe), leading and trailing clips, dense D / I; a record over many jobs; invalid bases on '-'
    strand rows; slices at the pool's edges.  Every case also through the block kernel (variant 0)."""
    from wgatools_amd import synth
    before = eng.get_param("expand_job_tiles")
    def chk(b, skip=None, variant=3):
        """symbol mode (no slices: nothing is left to the block kernel but giant tiles), then base mode"""
        check_pafpseudo(eng, b, 0, skip=skip, variant=variant)
        if variant == 3:
            assert eng.get_param("pseudo_stream_left_to_blocks") == 0
        check_pafpseudo(eng, b, 1, skip=skip, variant=variant)
    try:
        for jt in (1, 2, 8, 32):
            eng.set_param("expand_job_tiles", jt)
            rng = np.random.default_rng(100 + jt)
            b = synth.make_paf_batch(70 + jt, 9, 3000, 300_000)
            b = sprinkle_ops(rng, b, frac=0.03, codes=(3, 4, 4, 5, 6, 11))
            b = pseudo_clean(rng, b)
            chk(b, variant=3)
            assert eng.get_param("pseudo_stream_left_to_blocks") == 0
            tot = synth.class_sums(b["ops"] & 15, b["ops"] >> 4, b["op_off"])
            cols = (tot["mx"] + tot["d"]).astype(np.int64)
            skip = np.array([0, 5, 1023, 1024, 4097, int(cols[5]), int(cols[6]) - 1, int(cols[7]) // 2, 17], dtype=np.int64)
            skip = np.minimum(skip, cols)
            for v in (3, 0):
                chk(b, skip=skip, variant=v)
            # records 2 and 5: leftover bases behind the CIGAR, record 6 a slice that is too short (drain panics)
            u = dict(b)
            ql = b["q_src_len"].copy()
            ql[2] += 7
            ql[5] += 300
            ql[6] -= min(5, int(ql[6]))
            u["q_src_len"] = ql
            chk(u, skip=np.minimum(skip, 40), variant=3)
            assert eng.get_param("pseudo_stream_left_to_blocks") > 0
        eng.set_param("expand_job_tiles", 4)
        rng = np.random.default_rng(5)
        def dense(n):
            out = []
            for _ in range(n):
                out.append("%d=" % rng.integers(1, 4))
                out.append("%d%s" % (rng.integers(0, 3), "IDS"[int(rng.integers(0, 3))]))
            return "".join(out) + "7="
        cigars = ["5S" + dense(900) + "3S", "40=" + "1I" * 700 + "30=" + "2S1I" * 400 + "5=" + "1I" * 300, "1I" * 600 + "9=",
                  "5=" + dense(300) + "9000D3=" + dense(100) + "12000I4=2000S", dense(2500), "3=1I1D1I1D2=0I0D5=" * 40 + "1=",
                  "2000=" + "1D1I" * 500 + "1S" * 300 + "4D" + "1I" * 70 + "2000=", "7=" + "1I1S" * 160]
        strands = [0, 1, 0, 1, 1, 0, 1, 0]
        qs = [rand_seq(rng, pseudo_consumption(c)) for c in cigars]
        bt = batch_from_texts(eng, cigars, strands, [b"A"] * len(cigars), qs, pad=70)
        bt["q_pool"] = np.concatenate([np.frombuffer(b"N" * 64, np.uint8), bt["q_pool"], np.frombuffer(b"N" * 64, np.uint8)])
        bt["q_src_off"] = bt["q_src_off"] + np.uint64(64)
        for v in (3, 0):
            chk(bt, variant=v)
            chk(bt, skip=[3, 41, 0, 9100, 16, 0, 2300, 7], variant=v)
        chk(bt, variant=3)
        assert eng.get_param("pseudo_stream_left_to_blocks") == 0
        # more than 255 clip / insertion ops on ONE column inside one super-step (its event counters are bytes): ten at the end
        # of one intake of 256 ops, 255 at the start of the next, columns behind them
        cg = ["1=1X" * 123 + "1I" * 265 + "5000=", "7=", "1=1X" * 118 + "1S" * 270 + "3D1I" + "4000="]
        for st in ([0, 1, 1], [1, 0, 0]):
            bt = batch_from_texts(eng, cg, st, [b"A"] * 3, [rand_seq(rng, pseudo_consumption(c)) for c in cg], pad=80)
            bt["q_pool"] = np.concatenate([np.frombuffer(b"N" * 64, np.uint8), bt["q_pool"], np.frombuffer(b"N" * 64, np.uint8)])
            bt["q_src_off"] = bt["q_src_off"] + np.uint64(64)
            chk(bt, variant=3)
            assert eng.get_param("pseudo_stream_left_to_blocks") == 0
            chk(bt, skip=[250, 0, 236], variant=3)
        big = synth.make_paf_batch(12, 1, 40_000, 400_000, sigma=0.01)      # one record over ~40 tiles = 10 jobs
        big = pseudo_clean(rng, big)
        chk(big, variant=3)
        chk(big, skip=[30_011], variant=3)
        assert eng.get_param("pseudo_stream_left_to_blocks") == 0
        bad = pseudo_clean(rng, synth.make_paf_batch(13, 3, 3000, 100_000))   # invalid bases on '-' strand rows
        bad["strand_neg"][:] = 1
        qp = bad["q_pool"].copy()
        for r_, f in ((0, 3), (1, 2), (1, 5)):
            qp[int(bad["q_src_off"][r_] + bad["q_src_len"][r_] * f // 7)] = ord("R-x"[f % 3])
        bad["q_pool"] = qp
        for v in (3, 0):
            chk(bad, variant=v)
        # slices that begin at byte 0 and end at the last byte of the pool
        cig = ["700=3I900=2D650=", "1500=1X200=4D300=2S"]
        for st in ([0, 0], [1, 1]):
            chk(batch_from_texts(eng, cig, st, [b"A"] * 2, [rand_seq(rng, pseudo_consumption(c)) for c in cig], pad=0), variant=3)
            assert eng.get_param("pseudo_stream_left_to_blocks") > 0
    finally:
        eng.set_param("expand_job_tiles", before)


# ------------------------------------------------------------------------------------------------
# K3 MAF column pairs
# ------------------------------------------------------------------------------------------------
def binary_row_pairs(rng):
    """rows that are not text: bytes with bit 7 set (0xAD = '-' | 0x80, 0x80, 0xFF), zero bytes — the walks compare bytes, and
    their fast path for plain text must not be taken for these (one such byte among 1 024 columns, or many)"""
    out = []
    for L, hot in ((2100, 1), (1024, 40), (700, 700), (3000, 3)):
        t = bytearray(rand_seq(rng, L, b"ACGTacgt--N"))
        q = bytearray(rand_seq(rng, L, b"ACGTacgt--N"))
        for k in rng.integers(0, L, hot):
            which = int(rng.integers(0, 6))
            v = [0xAD, 0x80, 0xFF, 0x00, 0xC1, 0x2D][which]
            if rng.random() < 0.5:
                t[int(k)] = v
            else:
                q[int(k)] = v
            if rng.random() < 0.3:      # the same strange byte in both rows: equal
                t[int(k)] = q[int(k)] = v
        out.append((bytes(t), bytes(q)))
    return out


def check_maf_pair(eng, pairs, strands):
    """pairs: list of (t_row bytes, q_row bytes)"""
    n = len(pairs)
    buf, t_off, q_off, cols = bytearray(b"@@@"), [], [], []
    for t, q in pairs:
        t_off.append(len(buf))
        buf += t + b"@"
        q_off.append(len(buf))
        buf += q + b"@@"
        cols.append(min(len(t), len(q)))
    rows = eng.upload(np.frombuffer(bytes(buf), dtype=np.uint8))
    d_t, d_q = eng.upload(np.array(t_off, dtype=np.uint64)), eng.upload(np.array(q_off, dtype=np.uint64))
    d_c, d_s = eng.upload(np.array(cols, dtype=np.uint64)), eng.upload(np.array(strands, dtype=np.uint8))
    counts, run_cnt = eng.maf_pair_stat(n, rows, d_t, d_q, d_c, d_s)
    rc = run_cnt.numpy()
    run_off = eng.exclusive_scan_u64(n, run_cnt)
    ro = run_off.numpy()
    runs = eng.empty(int(ro[-1]) + 1, np.uint64).fill(0)
    counts, _ = eng.maf_pair_stat(n, rows, d_t, d_q, d_c, d_s, counts=counts, run_cnt=run_cnt,
                                  runs=runs, run_off=run_off)
    c, rr = counts.numpy(), runs.numpy()
    for i, (t, q) in enumerate(pairs):
        exp_counts, exp_txt = orc.parse_maf_seq_to_cigar(t, q, strands[i])
        assert tuple(int(x) for x in c[i]) == exp_counts, (i, exp_counts, c[i])
        mine = rr[int(ro[i]):int(ro[i + 1])]
        starts = (mine >> np.uint64(3)).astype(np.int64).tolist() + [cols[i]]
        txt = "".join("%d%s" % (starts[k + 1] - starts[k], "=IDX"[int(mine[k] & np.uint64(7))])
                      for k in range(len(mine)))
        assert txt == exp_txt, (i, txt[:80], exp_txt[:80])
        assert int(rc[i]) == len(mine)
    # K11: the same runs as maf2paf's cg:Z: text, and as packed ops through the chain kernel = maf2chain
    ne = int(ro[-1])
    tcnt = eng.maf_runs_cigar_text(n, ne, runs, run_off, d_c)
    toff = eng.exclusive_scan_u64(n, tcnt)
    to_ = toff.numpy()
    text = eng.empty(int(to_[-1]) + 8, np.uint8).fill(0x23)
    eng.maf_runs_cigar_text(n, ne, runs, run_off, d_c, out=text, out_off=toff)
    tx = text.numpy()
    ocnt = eng.maf_runs_ops(n, ne, runs, run_off, d_c)
    ooff = eng.exclusive_scan_u64(n, ocnt)
    oo = ooff.numpy()
    ops = eng.empty(int(oo[-1]) + 4, np.uint32).fill(0)
    eng.maf_runs_ops(n, ne, runs, run_off, d_c, out=ops, out_off=ooff)
    assert (oo == ro).all()                       # no run reaches 2^28 columns here
    batch = eng.make_batch_device(ops, ooff, d_s, n, int(oo[-1]))
    trim, nbytes, diag = eng.cigar_chain(batch)
    coff = eng.exclusive_scan_u64(n, nbytes)
    co = coff.numpy()
    ctext = eng.empty(int(co[-1]) + 8, np.uint8).fill(0x23)
    eng.cigar_chain(batch, out=ctext, out_off=coff)
    ct, tr = ctext.numpy(), trim.numpy()
    for i, (t, q) in enumerate(pairs):
        _, exp_txt = orc.parse_maf_seq_to_cigar(t, q, strands[i])
        assert tx[int(to_[i]):int(to_[i + 1])].tobytes().decode() == exp_txt, i
        assert tuple(int(tr[i][k]) for k in ("head_ins", "head_del", "tail_ins", "tail_del")) == \
            orc.parse_maf_seq_to_trim(t, q), i
        rec = orc.maf2chain_record("t", 10 ** 12, 0, 10 ** 11, "q", 10 ** 12, 0, 10 ** 11, strands[i], t, q, i)
        assert ct[int(co[i]):int(co[i + 1])].tobytes() == rec[rec.index(b"\n"):-2], (i, rec[:80])
    assert (tx[int(to_[-1]):] == 0x23).all() and (ct[int(co[-1]):] == 0x23).all()


def expected_split(length, code, cont):
    out, first = [], True
    while length:
        piece = min(length, (1 << 28) - 1)
        out.append((piece << 4) | (code if first else cont))
        length -= piece
        first = False
    return out


def check_runs_bridge_synthetic(eng, recs):
    """recs: per record a list of (length, class) runs, lengths up to and beyond 2^28 (no rows needed:
    the bridges only read the run list).  Ops = the PAF packer's split rule, text = '<len><op>'."""
    n = len(recs)
    runs, run_off, cols = [], [0], []
    for r in recs:
        pos = 0
        for ln, cls in r:
            runs.append((pos << 3) | cls)
            pos += ln
        run_off.append(len(runs))
        cols.append(pos)
    ne = len(runs)
    d_runs = eng.upload(np.array(runs + [0], dtype=np.uint64))
    d_roff = eng.upload(np.array(run_off, dtype=np.uint64))
    d_cols = eng.upload(np.array(cols, dtype=np.uint64))
    ocnt = eng.maf_runs_ops(n, ne, d_runs, d_roff, d_cols)
    ooff = eng.exclusive_scan_u64(n, ocnt)
    oo = ooff.numpy()
    ops = eng.empty(int(oo[-1]) + 4, np.uint32).fill(0xFFFFFFFF)
    eng.maf_runs_ops(n, ne, d_runs, d_roff, d_cols, out=ops, out_off=ooff)
    tcnt = eng.maf_runs_cigar_text(n, ne, d_runs, d_roff, d_cols)
    toff = eng.exclusive_scan_u64(n, tcnt)
    to_ = toff.numpy()
    text = eng.empty(int(to_[-1]) + 8, np.uint8).fill(0x23)
    eng.maf_runs_cigar_text(n, ne, d_runs, d_roff, d_cols, out=text, out_off=toff)
    o, tx = ops.numpy(), text.numpy()
    CODE, CONT = (7, 1, 2, 8), (7, 9, 10, 8)
    for i, r in enumerate(recs):
        want = [w for ln, cls in r for w in expected_split(ln, CODE[cls], CONT[cls])]
        assert o[int(oo[i]):int(oo[i + 1])].tolist() == want, i
        assert tx[int(to_[i]):int(to_[i + 1])].tobytes().decode() == "".join("%d%s" % (ln, "=IDX"[c]) for ln, c in r), i
    assert (o[int(oo[-1]):] == 0xFFFFFFFF).all() and (tx[int(to_[-1]):] == 0x23).all()


def check_bridge_blocks(eng, seed=3, n=24):
    """K11's fill: blocks of 256 elements inside one record go through an LDS buffer and out in 16-byte stores, blocks across
    a record border write directly.  Records of 150 - 900 data lines, the records' outputs placed with gaps of 0 - 37 units
    between them (every alignment of a stretch inside its 16-byte group): ops and text as expected, not a byte outside."""
    rng = np.random.default_rng(seed)
    L = (1 << 28) - 1
    recs = []
    for k in range(n):
        m = int(rng.integers(150, 900)) if k != 5 else 0
        r = [(int(rng.integers(0, 3000)), int(rng.integers(0, 3)) * int(rng.integers(0, 500)),
              int(rng.integers(0, 3)) * int(rng.integers(0, 70))) for _ in range(m)]
        if k == 7:
            r[100] = (3 * L + 2, 2 * L, L + 1)              # split lengths: more units than lines
        recs.append(r)
    flat = [v for r in recs for ln in r for v in ln]
    line_off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    ne = int(line_off[-1])
    d_lines = eng.upload(np.array(flat + [0, 0, 0], dtype=np.uint64))
    d_loff = eng.upload(line_off)
    for kind in ("ops", "text"):
        call = eng.chain_lines_ops if kind == "ops" else eng.chain_lines_cigar_text
        cnt = call(n, ne, d_lines, d_loff).numpy().astype(np.int64)
        gaps = rng.integers(0, 38, n)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum(cnt + gaps)
        off[:-1] += gaps.astype(np.uint64)                   # record i starts behind its gap
        total = int(off[-1]) + 40
        out = eng.empty(total, np.uint32 if kind == "ops" else np.uint8).fill(0xFF if kind == "ops" else 0x23)
        call(n, ne, d_lines, d_loff, out=out, out_off=eng.upload(off))
        h = out.numpy()
        guard = 0xFFFFFFFF if kind == "ops" else 0x23
        seen = np.zeros(total, dtype=bool)
        for i, r in enumerate(recs):
            a = int(off[i])
            if kind == "ops":
                want = [w for (sz, qd, td) in r for w in
                        expected_split(sz, 0, 0) + expected_split(td, 1, 9) + expected_split(qd, 2, 10)]
                assert len(want) == cnt[i] and h[a:a + len(want)].tolist() == want, (kind, i)
            else:
                want = "".join("%dM" % sz + ("%dI" % td if td else "") + ("%dD" % qd if qd else "") for sz, qd, td in r).encode()
                assert len(want) == cnt[i] and h[a:a + len(want)].tobytes() == want, (kind, i)
            seen[a:a + int(cnt[i])] = True
        assert (h[~seen] == guard).all(), kind


def check_elem_scan_reuse(eng):
    """K11: the count call's scan of the element sizes stays for the fill call.  A fill call without a count call in front
    computes its own; a second batch in the SAME device arrays (same pointers, same counts of records and lines, other
    lengths — other sizes per line) is counted anew and filled from the new scan"""
    def lines_of(seed):
        rng = np.random.default_rng(seed)
        recs = [[(int(rng.integers(0, 10 ** int(rng.integers(1, 9)))), int(rng.integers(0, 2)) * int(rng.integers(0, 5000)),
                  int(rng.integers(0, 2)) * int(rng.integers(0, 300))) for _ in range(m)] for m in (300, 1, 700, 256, 43)]
        return recs
    def text_of(r):
        return "".join("%dM" % sz + ("%dI" % td if td else "") + ("%dD" % qd if qd else "") for sz, qd, td in r).encode()
    n = 5
    a, b = lines_of(1), lines_of(2)
    line_off = np.cumsum([0] + [len(r) for r in a]).astype(np.uint64)
    ne = int(line_off[-1])
    flat = lambda recs: np.array([v for r in recs for ln in r for v in ln] + [0, 0, 0], dtype=np.uint64)
    d_lines, d_loff = eng.upload(flat(a)), eng.upload(line_off)
    def fill_and_check(recs, count_first):
        want = [text_of(r) for r in recs]
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(w) for w in want])
        if count_first:
            cnt = eng.chain_lines_cigar_text(n, ne, d_lines, d_loff).numpy().astype(np.int64)
            assert cnt.tolist() == [len(w) for w in want]
        out = eng.empty(int(off[-1]) + 16, np.uint8).fill(0x23)
        eng.chain_lines_cigar_text(n, ne, d_lines, d_loff, out=out, out_off=eng.upload(off))
        assert out.numpy()[:int(off[-1])].tobytes() == b"".join(want)
    fill_and_check(a, False)            # no count call in front
    fill_and_check(a, True)
    fill_and_check(a, False)            # the scan of the count call is still the right one
    eng.copy_into(d_lines, flat(b))     # the same arrays, another batch
    fill_and_check(b, True)
    eng.chain_lines_ops(n, ne, d_lines, d_loff)      # another entry point counts over the same arrays ...
    fill_and_check(b, False)                          # ... and the text fill does not take its scan
    # a count call, then the arrays' contents replaced through the library, then a fill call WITHOUT a count call: the
    # scan of the first batch must not be taken (it was, silently)
    cnt = eng.chain_lines_cigar_text(n, ne, d_lines, d_loff).numpy().astype(np.int64)
    assert cnt.tolist() == [len(text_of(r)) for r in b]
    eng.copy_into(d_lines, flat(a))
    fill_and_check(a, False)
    # ... and a count-to-fill scan is one-shot: the second fill call after one count call computes its own
    fill_and_check(a, True)
    eng.copy_into(d_lines, flat(b))
    fill_and_check(b, False)


def check_chain_lines(eng, recs, strands, seqs=None):
    """recs: per record a list of (size, query_diff, target_diff) data lines.  ops + K1 == the counts of
    parse_chain_to_cigar, text == its CIGAR, and (seqs given: per record (t_seq, q_seq)) ops + K2 ==
    parse_chain_to_insert."""
    n = len(recs)
    flat = [v for r in recs for ln in r for v in ln]
    line_off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    ne = int(line_off[-1])
    d_lines = eng.upload(np.array(flat + [0, 0, 0], dtype=np.uint64))
    d_loff = eng.upload(line_off)
    ocnt = eng.chain_lines_ops(n, ne, d_lines, d_loff)
    ooff = eng.exclusive_scan_u64(n, ocnt)
    oo = ooff.numpy()
    ops = eng.empty(int(oo[-1]) + 4, np.uint32).fill(0xFFFFFFFF)
    eng.chain_lines_ops(n, ne, d_lines, d_loff, out=ops, out_off=ooff)
    tcnt = eng.chain_lines_cigar_text(n, ne, d_lines, d_loff)
    toff = eng.exclusive_scan_u64(n, tcnt)
    to_ = toff.numpy()
    text = eng.empty(int(to_[-1]) + 8, np.uint8).fill(0x23)
    eng.chain_lines_cigar_text(n, ne, d_lines, d_loff, out=text, out_off=toff)
    o, tx = ops.numpy(), text.numpy()
    strand = np.array(strands, dtype=np.uint8)
    batch = eng.make_batch_device(ops, ooff, eng.upload(strand), n, int(oo[-1]))
    counts, diag, _ = eng.cigar_stat(batch)
    c = counts.numpy()
    for i, r in enumerate(recs):
        want = [w for (sz, qd, td) in r for w in
                expected_split(sz, 0, 0) + expected_split(td, 1, 9) + expected_split(qd, 2, 10)]
        assert o[int(oo[i]):int(oo[i + 1])].tolist() == want, i
        exp_counts, exp_txt = orc.parse_chain_to_cigar(r, strands[i])
        assert tx[int(to_[i]):int(to_[i + 1])].tobytes().decode() == exp_txt, (i, exp_txt[:80])
        assert tuple(int(x) for x in c[i]) == exp_counts, (i, exp_counts, c[i])
    assert (o[int(oo[-1]):] == 0xFFFFFFFF).all() and (tx[int(to_[-1]):] == 0x23).all()
    if seqs is None:
        return
    def pool(ss):
        offs, buf = [], bytearray(b"N" * 16)
        for x in ss:
            offs.append(len(buf))
            buf += x
        buf += b"N" * 16
        return np.frombuffer(bytes(buf), dtype=np.uint8).copy(), np.array(offs, dtype=np.uint64)
    tp, to = pool([t for t, _ in seqs])
    qp, qo = pool([q for _, q in seqs])
    b = dict(ops=o[:int(oo[-1])].copy(), op_off=oo.copy(), strand_neg=strand, t_pool=tp, q_pool=qp, t_src_off=to,
             q_src_off=qo, t_src_len=np.array([len(t) for t, _ in seqs], dtype=np.uint64),
             q_src_len=np.array([len(q) for _, q in seqs], dtype=np.uint64))
    r = run_paf2maf(eng, b)
    for i, (t, q) in enumerate(seqs):
        qq = orc.reverse_complement(q) if strands[i] else q
        try:
            et, eq = orc.parse_chain_to_insert(recs[i], t, qq)
        except orc.OracleError as e:
            assert e.kind == 6 and int(r["diag"][i]["panic_op_idx"]) != NONE, (i, r["diag"][i])
            continue
        assert int(r["diag"][i]["panic_op_idx"]) == NONE, (i, r["diag"][i])
        a, bq = int(r["t_row_off"][i]), int(r["q_row_off"][i])
        assert r["out"][a:a + len(et)].tobytes() == et and r["out"][bq:bq + len(eq)].tobytes() == eq, i


# ------------------------------------------------------------------------------------------------
# K4 MAF column runs for `call`
# ------------------------------------------------------------------------------------------------
def check_maf_call_runs(eng, pairs):
    """runs of equal cigar_cat_ext_caller class (cigar.rs:314-328, grouped as caller.rs:444-446)
    with the non-gap target / query characters before each run (what create_chunk_record,
    caller.rs:221-265, and the offsets of call_within_var count)."""
    n = len(pairs)
    buf, t_off, q_off, cols = bytearray(b"@"), [], [], []
    for t, q in pairs:
        t_off.append(len(buf))
        buf += t + b"@@@"
        q_off.append(len(buf))
        buf += q + b"@"
        cols.append(min(len(t), len(q)))
    rows = eng.upload(np.frombuffer(bytes(buf), dtype=np.uint8))
    d_t, d_q = eng.upload(np.array(t_off, dtype=np.uint64)), eng.upload(np.array(q_off, dtype=np.uint64))
    d_c = eng.upload(np.array(cols, dtype=np.uint64))
    run_cnt = eng.maf_call_runs(n, rows, d_t, d_q, d_c)
    run_off = eng.exclusive_scan_u64(n, run_cnt)
    ro = run_off.numpy()
    runs = eng.empty(3 * int(ro[-1]) + 3, np.uint64).fill(0)
    eng.maf_call_runs(n, rows, d_t, d_q, d_c, run_cnt=run_cnt, runs=runs, run_off=run_off)
    rr = runs.numpy()
    for i, (t, q) in enumerate(pairs):
        L = cols[i]
        ta, qa = np.frombuffer(t, dtype=np.uint8)[:L], np.frombuffer(q, dtype=np.uint8)[:L]
        tg, qg = ta == 45, qa == 45
        cls = np.where(tg & qg, 4, np.where(tg, 1, np.where(qg, 2, np.where(ta == qa, 0, 3))))
        starts = np.flatnonzero(np.concatenate([[True], cls[1:] != cls[:-1]])) if L else np.zeros(0, np.int64)
        tb = np.concatenate([[0], np.cumsum(~tg)])[starts] if L else starts
        qb = np.concatenate([[0], np.cumsum(~qg)])[starts] if L else starts
        mine = rr[3 * int(ro[i]):3 * int(ro[i + 1])].reshape(-1, 3)
        assert len(mine) == len(starts), (i, len(mine), len(starts))
        assert ((mine[:, 0] >> np.uint64(3)).astype(np.int64) == starts).all(), i
        assert ((mine[:, 0] & np.uint64(7)).astype(np.int64) == cls[starts]).all(), i
        assert (mine[:, 1].astype(np.int64) == tb).all(), i
        assert (mine[:, 2].astype(np.int64) == qb).all(), i
        # ... and against the ORACLE: the variants call_within_var (caller.rs:388-608, orc_call_within_var) reports for
        # this pair as one chunk with -s -l0 are exactly what the run list implies — a SNP per column of an X run, an
        # INS / DEL per target- / query-gap run that follows an = or X run (both-gap runs in between do not count),
        # anchored at the target base before it
        is_text = L and int(ta.max()) < 0x80 and int(qa.max()) < 0x80 and int(ta.min()) > 0 and int(qa.min()) > 0
        if L and not (tg & qg).all() and is_text:   # the oracle wrapper hands the VCF back as text
            t_start = 1000 + 7 * i
            vcf = orc.call_within_var("chrT", "qry", t, q, t_start, t_start + int((~tg).sum()), 50, 50 + int((~qg).sum()),
                                      False, True, 0, False)
            want = [(int(f[1]), "INS" if "SVTYPE=INS" in f[7] else "DEL" if "SVTYPE=DEL" in f[7] else "SNP",
                     int(f[7].split("SVLEN=")[1].split(";")[0]) if "SVLEN=" in f[7] else 1)
                    for f in (ln.split("\t") for ln in vcf.splitlines())]
            got, after_m = [], False   # after_m (caller.rs:453): set by = / X groups, cleared by I / D, kept by W
            ends = list(mine[1:, 0] >> np.uint64(3)) + [L]
            for k in range(len(mine)):
                s0, c = int(mine[k, 0] >> np.uint64(3)), int(mine[k, 0] & np.uint64(7))
                ln, tbk = int(ends[k]) - s0, int(mine[k, 1])
                if c == 3:
                    got += [(t_start + tbk + j + 1, "SNP", 1) for j in range(ln)]
                elif c in (1, 2) and after_m:
                    got.append((t_start + tbk, "INS" if c == 1 else "DEL", ln))
                after_m = True if c in (0, 3) else False if c in (1, 2) else after_m
            d = next((k for k in range(min(len(got), len(want))) if got[k] != want[k]), min(len(got), len(want)))
            assert got == want, (i, d, got[max(0, d - 2):d + 3], want[max(0, d - 2):d + 3], t[:60], q[:60])


# ------------------------------------------------------------------------------------------------
# K7 PAF call op walk
# ------------------------------------------------------------------------------------------------
def expected_paf_call_events(ops, svlen, snp):
    """op-serial restatement of the fold of call_within_var_paf (caller.rs:664-819) over packed ops"""
    ev, t, q, after_m, k = [], 0, 0, False, 0
    n = len(ops)
    while k < n:
        code, ln = int(ops[k]) & 15, int(ops[k]) >> 4
        if code in (0, 7):
            t += ln; q += ln; after_m = True; k += 1
        elif code == 8:
            if snp:
                ev.append((k, t, q))
            t += ln; q += ln; after_m = True; k += 1
        elif code in (1, 2):
            j, tot = k + 1, ln
            while j < n and (int(ops[j]) & 15) in (9, 10):
                tot += int(ops[j]) >> 4
                j += 1
            if after_m and (ln > svlen or j > k + 1):     # the host applies `tot > svlen` to split ops
                ev.append((k, t, q))
            if code == 1:
                q += tot
            else:
                t += tot
            after_m = False
            k = j
        else:
            break
    return ev


def check_paf_call_events(eng, ops, op_off, svlen, snp):
    n = len(op_off) - 1
    batch = eng.make_batch(ops, op_off, np.zeros(n, dtype=np.uint8))
    cnt = eng.paf_call_events(batch, svlen, snp)
    off = eng.exclusive_scan_u64(n, cnt)
    o = off.numpy()
    ev = eng.empty(3 * int(o[-1]) + 3, np.uint64).fill(0)
    eng.paf_call_events(batch, svlen, snp, ev_cnt=cnt, ev=ev, ev_off=off)
    e = ev.numpy()
    rng = np.random.default_rng(17)
    for i in range(n):
        rops = ops[int(op_off[i]):int(op_off[i + 1])]
        want = expected_paf_call_events(rops, svlen, snp)
        got = [tuple(int(x) for x in e[3 * k:3 * k + 3]) for k in range(int(o[i]), int(o[i + 1]))]
        assert got == want, (i, got[:5], want[:5])
        # ... and against the ORACLE (orc_call_within_var_paf, caller.rs:610-822): the event list implies exactly the
        # VCF rows it writes for the record — one SNP per base of an X op, one INS / DEL per event op, anchored at the
        # target base before it.  Records with split (>= 2^28) indels are left to the python restatement above.
        if len(rops) and not ((rops & 15) >= 9).any() and len(rops) < 4000:
            t_cons = int(sum(int(w) >> 4 for w in rops if (int(w) & 15) in (0, 2, 7, 8)))
            q_cons = int(sum(int(w) >> 4 for w in rops if (int(w) & 15) in (0, 1, 7, 8)))
            t_seq, q_seq = rand_seq(rng, t_cons + 1, b"ACGT"), rand_seq(rng, q_cons + 1, b"ACGT")
            t_start = 500 + 3 * i
            vcf = orc.call_within_var_paf("tchr", "qchr", text_any(rops), t_seq, q_seq, t_start, t_start + t_cons, 70,
                                          70 + q_cons, False, snp, svlen)
            want_rows = [(int(f[1]), "INS" if "SVTYPE=INS" in f[7] else "DEL" if "SVTYPE=DEL" in f[7] else "SNP")
                         for f in (ln.split("\t") for ln in vcf.splitlines())]
            got_rows = []
            for k, t, q in got:
                code, ln = int(rops[k]) & 15, int(rops[k]) >> 4
                if code == 8:
                    got_rows += [(t_start + t + j + 1, "SNP") for j in range(ln)]
                else:
                    got_rows.append((t_start + t, "INS" if code == 1 else "DEL"))
            assert got_rows == want_rows, (i, got_rows[:6], want_rows[:6])
    return int(o[-1])


def long_record_ops(seed, n_long, long_ops, bad_in=None, shorts=1):
    """records mixed short / long: every `long` one holds `long_ops` ops and is followed by `shorts` batches of 3 short
    ones (few of them: every record goes through the piece kernels; many: the short ones stay with the one-wave kernel);
    bad_in = (record, op index) gets an op the walks stop at.  Piece boundaries (multiples of 256) are made nasty: an indel right behind a boundary, a continuation piece
    of a split indel starting a piece, X ops at the cut."""
    from wgatools_amd import synth
    rng = np.random.default_rng(seed)
    L = (1 << 28) - 1
    recs = []
    for k in range(n_long):
        ops, _, _, _ = synth.make_ops(rng, 1, long_ops, sigma=0.01, min_ops=long_ops, max_ops=long_ops)
        ops = ops.copy()
        for cut in range(256, len(ops) - 2, 256):
            kind = (cut // 256 + k) % 4
            if kind == 0:
                ops[cut - 1] = (7 << 4) | 7          # '=' then an indel that opens the piece
                ops[cut] = (60 << 4) | (1 + (cut // 256) % 2)
            elif kind == 1:
                ops[cut - 1] = (L << 4) | 1          # a split I whose continuation starts the piece
                ops[cut] = (5 << 4) | 9
                ops[cut - 2] = (3 << 4) | 7
            elif kind == 2:
                ops[cut - 1] = (2 << 4) | 8          # X | X across the cut
                ops[cut] = (1 << 4) | 8
        recs.append(ops)
        for _ in range(shorts):
            short, _, _, _ = synth.make_ops(rng, 3, 40)
            recs.append(short)
    if bad_in is not None:
        recs[bad_in[0]][bad_in[1]] = (4 << 4) | 3     # N
    ops = np.concatenate(recs).astype(np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    return ops, off


def check_paf_call_long_records(eng, mops=2):
    """K7 with records cut into pieces: small knobs (pieces of 256 ops) on nasty cuts and a stop inside a middle piece,
    then one record of `mops` million ops with the product's piece size"""
    eng.set_param("op_long_ops", 600)
    eng.set_param("op_piece_ops", 256)
    try:
        ops, off = long_record_ops(5, 3, 2300)
        for svlen, snp in ((0, True), (8, False), (50, True)):
            check_paf_call_events(eng, ops, off, svlen, snp)
        ops, off = long_record_ops(6, 3, 1500, bad_in=(2, 700))
        check_paf_call_events(eng, ops, off, 3, True)
        ops, off = long_record_ops(9, 2, 2300, shorts=12)      # mostly short records: two kernels share the batch
        check_paf_call_events(eng, ops, off, 4, True)
    finally:
        eng.set_param("op_long_ops", 16384)
        eng.set_param("op_piece_ops", 8192)
    if mops:
        ops, off = long_record_ops(7, 1, mops * 1_000_000)
        check_paf_call_events(eng, ops, off, 20, True)


# ------------------------------------------------------------------------------------------------
# K8 device tokeniser (and the host packer) vs the oracle's token stream
# ------------------------------------------------------------------------------------------------
TOKENISER_EDGE_TEXTS = [
    b"", b"5", b"M", b"5M", b"10=3I", b"12", b"3M4", b"3MM2I", b"3M2II", b"I", b"4=I3M", b"0M", b"00012=",
    b"0000000000000000000000000000000000000000000005M2I",       # 40+ leading zeros: beyond the history window
    b"18446744073709551616M", b"99999999999999999999M3I",         # u64 overflow (20 digits)
    b"0000000000000000000000012345M",                              # > 19 digits, small value
    b"268435455M", b"268435456M", b"536870911I4=", b"1073741824D",  # split lengths (2^28 - 1 is the largest op)
    "3\u00e9".encode(), "3\u00e95M".encode(), b"3\xc3", b"4\xe2\x82\xac2M", b"2M\xff\xfe3I", b"7 M", b"7\tM3I",
    b"5B3z", b"10=2X" * 100 + b"4", b"10=2X" * 100 + b"MM", b"1=" * 700,  # > 1 KiB: errors in a later chunk
    b"123456789=" * 120 + b"5I", b"9" * 15 + b"M" + b"1=" * 520,
]


PACK_CODES = {ord(c): k for k, c in enumerate("MIDNSHP=X")}
OP_MAX_LEN = (1 << 28) - 1


def oracle_packed(text):
    """the packed u32 stream the boundary defines (include/wga_hip.h: len << 4 | code, lengths beyond 2^28 - 1 split,
    later pieces of an I / D marked as continuations) for the tokens the ORACLE yields (orc_tokenise, cigar.rs:43-75),
    its error kind as a wga_rec_err and the quoted token span"""
    toks, kind, span = orc.tokenise(text)
    ops = []
    for ln, ch in toks:
        code = PACK_CODES.get(ch, 11)
        cont = 9 if code == 1 else 10 if code == 2 else code
        first = True
        while True:
            piece = min(ln, OP_MAX_LEN)
            ops.append((piece << 4) | (code if first else cont))
            ln -= piece
            first = False
            if ln == 0:
                break
    return np.array(ops, dtype=np.uint32), kind, span


def check_tokeniser(eng, texts):
    """count pass + scan + fill pass must reproduce the oracle's token stream (packed as the boundary defines) for
    every record, its error kind and the token the message quotes — and so must the host packer"""
    n = len(texts)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(t) for t in texts], dtype=np.uint64)
    blob = np.frombuffer(b"".join(texts) + b"0" * 16, dtype=np.uint8)   # a few bytes of slack behind the last text
    d_text, d_off = eng.upload(blob), eng.upload(offs)
    cnt, err = eng.cigar_tokenise(n, d_text, d_off)
    op_off = eng.exclusive_scan_u64(n, cnt)
    oo = op_off.numpy()
    ops = eng.empty(int(oo[-1]) + 1, np.uint32).fill(0xEE)
    eng.cigar_tokenise(n, d_text, d_off, op_cnt=cnt, err=err, ops=ops, op_off=op_off)
    c, e, o = cnt.numpy(), err.numpy(), ops.numpy()
    for i, t in enumerate(texts):
        want_ops, want_err, (eo, el) = oracle_packed(t)
        h_ops, h_err, h_span = eng.pack_cigar(t)       # the host packer of the same library
        assert (h_ops == want_ops).all() and h_err == want_err, (i, t[:40], h_err, want_err)
        if want_err not in (0, 6):
            assert tuple(h_span) == (eo, el), (i, t[:40], h_span, eo, el)
        assert int(c[i]) == len(want_ops), (i, t[:40], int(c[i]), len(want_ops))
        got = o[int(oo[i]):int(oo[i + 1])]
        assert (got == want_ops).all(), (i, t[:40], got[:8], want_ops[:8])
        assert int(e[i]["err"]) == want_err, (i, t[:40], int(e[i]["err"]), want_err)
        if want_err not in (0, 6):
            assert (int(e[i]["tok_off"]), int(e[i]["tok_len"])) == (eo, el), (i, t[:40], e[i], eo, el)
    assert int(o[int(oo[-1])]) == 0xEEEEEEEE



# ------------------------------------------------------------------------------------------------
# K9 pafcov BED text
# ------------------------------------------------------------------------------------------------
def check_pafcov_format(eng, name, cov, p0):
    """lines "<name>\\t<pos>\\t<pos+1>\\t<count>\\n" exactly as pafcov.rs:56-60 prints them"""
    cov = np.asarray(cov, dtype=np.int32)
    n = len(cov)
    d_name = eng.upload(np.frombuffer(name or b"\0", dtype=np.uint8)[: max(1, len(name))])
    d_name.shape = (len(name),)
    d_cov = eng.upload(cov if n else np.zeros(1, np.int32))
    off = eng.pafcov_format(d_name, d_cov, p0, n)
    o = off.numpy()
    want = b"".join(b"%s\t%d\t%d\t%d\n" % (name, p0 + i, p0 + i + 1, int(np.uint32(cov[i]))) for i in range(n))
    assert int(o[n]) - int(o[0]) == len(want), (int(o[n]), len(want))
    out = eng.empty(len(want) + 8, np.uint8).fill(0x23)
    eng.pafcov_format(d_name, d_cov, p0, n, line_off=off, out=out)
    got = out.numpy()
    assert got[: len(want)].tobytes() == want
    assert (got[len(want):] == 0x23).all()



# ------------------------------------------------------------------------------------------------
# K10 paf2chain data lines + header trim
# ------------------------------------------------------------------------------------------------
def check_cigar_chain(eng, ops, op_off):
    """per record: trims = parse_cigar_to_trim, text = the data lines of parse_cigar_to_chain (between the
    header and the closing blank line of the oracle's record text)"""
    n = len(op_off) - 1
    batch = eng.make_batch(ops, op_off, np.zeros(n, dtype=np.uint8))
    trim, nbytes, diag = eng.cigar_chain(batch)
    tr, nb, dg = trim.numpy(), nbytes.numpy(), diag.numpy()
    off = eng.exclusive_scan_u64(n, nbytes)
    oo = off.numpy()
    out = eng.empty(int(oo[-1]) + 8, np.uint8).fill(0x23)
    eng.cigar_chain(batch, out=out, out_off=off)
    o = out.numpy()
    def text_merged(sl):   # a head op and its continuation pieces (codes 9 / 10) are ONE op of the text
        toks = []
        for w in sl.tolist():
            c, ln = w & 15, w >> 4
            if c in (9, 10) and toks:
                toks[-1][0] += ln
            else:
                toks.append([ln, synth.OP_CHARS[c] if c < 9 else "B"])
        return "".join("%d%s" % (ln, ch) for ln, ch in toks)
    for i in range(n):
        cg = "cg:Z:" + text_merged(ops[int(op_off[i]):int(op_off[i + 1])])
        try:
            want_tr = orc.parse_cigar_to_trim(cg)
        except orc.OracleError as e:
            if e.kind == 2:                      # CigarOpInvalid: first op outside M = X I D
                assert int(dg[i]["bad_op_idx"]) != NONE, (i, cg[:60])
                codes = (ops[int(op_off[i]):int(op_off[i + 1])] & 15)
                firstbad = int(np.flatnonzero(~np.isin(codes, [0, 7, 8, 1, 2, 9, 10]))[0])
                assert int(dg[i]["bad_op_idx"]) == firstbad
            continue                             # empty CIGAR: the host refuses it before the kernel
        assert int(dg[i]["bad_op_idx"]) == NONE, (i, cg[:60])
        assert tuple(int(tr[i][k]) for k in ("head_ins", "head_del", "tail_ins", "tail_del")) == want_tr, (i, cg[:80], tr[i], want_tr)
        rec = orc.paf2chain_record("q", 10 ** 12, 0, 10 ** 11, 0, "t", 10 ** 12, 0, 10 ** 11, cg, 0)
        want = rec[rec.index(b"\n"):-2]
        got = o[int(oo[i]):int(oo[i + 1])].tobytes()
        assert int(nb[i]) == len(want), (i, cg[:80], int(nb[i]), len(want))
        assert got == want, (i, cg[:80], got[:60], want[:60])
    assert (o[int(oo[-1]):] == 0x23).all()



def check_cigar_chain_long_records(eng, mops=2):
    """K10 with records cut into pieces where a line is certain: small knobs (pieces of 256 ops) on nasty cuts, records
    without any cut (their pieces are empty), zero-length ops where a cut would be, a stop inside a middle piece, sums past
    2^32 inside one piece (the record goes through the serial walk), then one record of `mops` million ops with the
    product's piece size"""
    L = (1 << 28) - 1
    mk = lambda p: np.array([(int(ln) << 4) | int(c) for c, ln in p], dtype=np.uint32)
    eng.set_param("op_long_ops", 600)
    eng.set_param("op_piece_ops", 256)
    try:
        for seed in (5, 8):
            ops, off = long_record_ops(seed, 3, 2300)
            check_cigar_chain(eng, ops, off)
        ops, off = long_record_ops(6, 3, 1500, bad_in=(2, 700))
        check_cigar_chain(eng, ops, off)
        ops, off = long_record_ops(9, 2, 2300, shorts=12)      # mostly short records: two kernels share the batch
        check_cigar_chain(eng, ops, off)
        recs = [
            mk([(7, 2), (8, 1)] * 900),                                           # no indel: one piece, the others empty
            mk([(7, 2), (1, 0), (7, 1)] * 500),                                   # zero-length indels: never a cut
            mk([(7, 0), (1, 2), (7, 1)] * 500),                                   # zero-length blocks in front of the indel
            mk([(1, 3)] * 700 + [(7, 4), (2, 2), (7, 1)] * 300),                  # the head trim is longer than two pieces
            mk([(7, 5), (2, 1)] * 400 + [(1, 2)] * 700),                          # a tail longer than two pieces
            mk([(7, 3), (1, 1)] * 200 + [(7, L)] * 20 + [(1, 1), (7, 3)] * 200),  # 2^32 inside a middle piece
            mk([(7, 1), (1, 1), (2, 1)] * 400 + [(7, 9)]),                        # I D groups: no M indel M pattern at all
            mk([(7, 1), (1, 1), (2, 1)] * 300 + [(7, 2), (2, 5), (7, 9)] + [(8, 1)] * 400),  # one cut, late
        ]
        ops = np.concatenate(recs)
        off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
        check_cigar_chain(eng, ops, off)
    finally:
        eng.set_param("op_long_ops", 16384)
        eng.set_param("op_piece_ops", 8192)
    if mops:
        ops, off = long_record_ops(7, 1, mops * 1_000_000)
        check_cigar_chain(eng, ops, off)


def check_piece_table_rebuild(eng):
    """the count call of K7 / K10 / K12 leaves its piece table for the fill call; a fill call that does not find it (another
    walk ran in between) builds it again: both ways give the same output"""
    eng.set_param("op_long_ops", 600)
    eng.set_param("op_piece_ops", 256)
    try:
        ops, off = long_record_ops(11, 3, 2300)
        n = len(off) - 1
        batch = eng.make_batch(ops, off, np.zeros(n, dtype=np.uint8))
        zeros = eng.upload(np.zeros(n, dtype=np.uint64))
        def k7(between):
            cnt = eng.paf_call_events(batch, 5, True)
            between()
            eo = eng.exclusive_scan_u64(n, cnt)
            ev = eng.empty(3 * int(eo.numpy()[-1]) + 3, np.uint64).fill(0)
            eng.paf_call_events(batch, 5, True, ev_cnt=cnt, ev=ev, ev_off=eo)
            return ev.numpy().copy()
        def k10(between):
            _, nb, _ = eng.cigar_chain(batch)
            between()
            oo = eng.exclusive_scan_u64(n, nb)
            out = eng.empty(int(oo.numpy()[-1]) + 8, np.uint8).fill(0)
            eng.cigar_chain(batch, out=out, out_off=oo)
            return out.numpy().copy()
        def k12(between):
            cnt = eng.cigar_dotplot(batch, 0, zeros, zeros)
            between()
            so = eng.exclusive_scan_u64(n, cnt)
            sg = eng.empty(5 * int(so.numpy()[-1]) + 5, np.uint64).fill(0)
            eng.cigar_dotplot(batch, 0, zeros, zeros, segs=sg, seg_off=so)
            return sg.numpy().copy()
        walks = (k7, k10, k12)
        for x, w in enumerate(walks):
            kept = w(lambda: None)
            other = walks[(x + 1) % 3]
            rebuilt = w(lambda: other(lambda: None))
            assert kept.size > 100 and (kept == rebuilt).all(), x
    finally:
        eng.set_param("op_long_ops", 16384)
        eng.set_param("op_piece_ops", 8192)


def chain_stress_records(seed, n=40, max_ops=2600):
    """records that exercise the step structure of K10 (512-op steps, 8 ops per lane, 64 lines per round): several
    steps, steps without a raise, a raise on every other op, leading / trailing indel runs longer than a step, ten-digit
    values, open sums that pass 2^32 (the serial path), continuation pieces"""
    rng = np.random.default_rng(seed)
    L = (1 << 28) - 1
    mk = lambda p: [(int(ln) << 4) | int(c) for c, ln in p]
    recs = []
    recs.append(mk([(7, 1), (1, 1)] * 1300 + [(7, 3)]))                       # a raise on every other op, 6 steps
    recs.append(mk([(7, 2), (8, 1)] * 700 + [(1, 5), (7, 9)]))                # three steps without a raise, then one
    recs.append(mk([(1, 3)] * 600 + [(2, 1)] * 30 + [(7, 4), (2, 2), (7, 1)]))  # the first M-like op in step 2
    recs.append(mk([(7, 5)] + [(2, 7)] * 1100))                               # a tail longer than two steps
    recs.append(mk([(7, 5)] + [(1, 2), (9, 3)] * 40 + [(2, 9)] * 70))         # tail I behind 64+ D ops
    recs.append(mk([(7, L), (8, L)] * 5 + [(1, L), (7, 1)]))                  # ten digits
    recs.append(mk([(7, L)] * 20 + [(1, 1), (7, L)]))                         # 2^32 passed inside one block: serial
    recs.append(mk([(7, 4)] * 511 + [(1, 1)] + [(7, 2)] * 3))                 # the raise is op 0 of a step
    recs.append(mk([(7, 4)] * 510 + [(1, 1), (2, 1), (7, 2), (1, 1)]))        # group across the step's end, I at the end
    recs.append(mk([(2, 1)] * 512 + [(7, 1)]))                                # dropped line is the 2nd step's first
    recs.append(mk([(7, 1), (2, 1)] * 256))                                   # exactly one step, ends in D
    recs.append(mk([(7, 1), (2, 1)] * 256 + [(7, 1)]))
    # zero-length ops and ops outside M = X I D inside later steps: those steps run the reference's loop as it stands
    recs.append(mk([(7, 3), (1, 2)] * 400 + [(7, 0), (2, 3), (7, 4), (1, 0), (7, 2), (2, 0), (1, 1), (7, 0), (7, 5)] + [(7, 1), (2, 2)] * 300))
    recs.append(mk([(1, 0)] * 3 + [(7, 0), (2, 2), (7, 3)] + [(8, 1), (1, 1)] * 600))          # 0I 0= lead the record
    recs.append(mk([(2, 4)] * 520 + [(7, 0), (1, 2), (7, 3), (2, 1), (7, 1)]))                  # the first M-like op is 0=
    recs.append(mk([(7, 2), (1, 1)] * 700 + [(3, 5)] + [(7, 2), (1, 1)] * 100))                 # N in the third step
    recs.append(mk([(7, 2), (1, 1)] * 300 + [(2, 0), (4, 1), (7, 1)]))                          # S behind a 0D, last step
    recs.append(mk([(7, 2), (2, 0)] * 600 + [(1, 3)]))                                          # every indel is empty
    for _ in range(n):
        k = int(rng.integers(1, max_ops))
        style = int(rng.integers(0, 4))
        if style == 0:      # mostly M-like, few indels
            codes = rng.choice([7, 8, 0, 1, 2], size=k, p=[0.55, 0.3, 0.05, 0.05, 0.05])
        elif style == 1:    # indel heavy
            codes = rng.choice([7, 8, 1, 2, 9, 10], size=k, p=[0.3, 0.1, 0.25, 0.25, 0.05, 0.05])
        elif style == 2:    # long runs of one kind
            codes = np.repeat(rng.choice([7, 1, 2, 8], size=k // 50 + 1), 50)[:k]
        else:
            codes = rng.choice([7, 1, 2], size=k)
        lens = rng.integers(1, 5000, size=k)
        big = rng.random(k) < 0.01
        lens = np.where(big, rng.integers(1, L, size=k), lens)
        pairs = []
        for c, ln in zip(codes.tolist(), lens.tolist()):
            if c in (9, 10):            # a continuation piece follows an op of its own kind (the packer's invariant)
                c -= 8
                pairs.append((c, ln))
                c += 8
            pairs.append((c, ln))
        recs.append(mk(pairs))
    ops = np.array([o for r in recs for o in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    return ops, off


# ------------------------------------------------------------------------------------------------
# a second, linear-time expectation for long records (the C oracle's insert_str is quadratic)
# ------------------------------------------------------------------------------------------------
def fast_expected_rows(ops, t_seq, q_seq, neg):
    """numpy gather formulation of converter.rs:228-235 for clean records (ops in M = X I D and
    slices exactly as long as the CIGAR consumes).  Cross-checked against the C oracle in
    test_emu_parity.py::test_fast_expected_matches_oracle."""
    code = (ops & 15).astype(np.int64)
    ln = (ops >> 4).astype(np.int64)
    is_i = (code == 1) | (code == 9)
    is_d = (code == 2) | (code == 10)
    t_adv = np.where(is_i, 0, ln)
    q_adv = np.where(is_d, 0, ln)
    t_before = np.concatenate([[0], np.cumsum(t_adv)[:-1]])
    q_before = np.concatenate([[0], np.cumsum(q_adv)[:-1]])
    col_before = np.concatenate([[0], np.cumsum(ln)[:-1]])
    L = int(ln.sum())
    op_of_col = np.repeat(np.arange(len(ops)), ln)
    within = np.arange(L) - col_before[op_of_col]
    t = np.frombuffer(t_seq, dtype=np.uint8)
    q = np.frombuffer(q_seq, dtype=np.uint8)
    if neg:
        comp = np.arange(256, dtype=np.uint8)
        for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
            comp[a] = b
        q = comp[q[::-1]]
    ti = np.minimum(t_before[op_of_col] + within, max(len(t) - 1, 0))
    qi = np.minimum(q_before[op_of_col] + within, max(len(q) - 1, 0))
    t_row = np.where(is_i[op_of_col], ord("-"), t[ti] if len(t) else ord("-")).astype(np.uint8)
    q_row = np.where(is_d[op_of_col], ord("-"), q[qi] if len(q) else ord("-")).astype(np.uint8)
    return t_row.tobytes(), q_row.tobytes()


# ------------------------------------------------------------------------------------------------
# K12 dotplot base-level segments
# ------------------------------------------------------------------------------------------------
def check_dotplot(eng, ops, op_off, strands, cutoff, seed=3):
    """per record: the segment list of parse_cigar_to_base_plotdata (cigar.rs:917-952)"""
    n = len(op_off) - 1
    rng = np.random.default_rng(seed)
    ts = rng.integers(0, 10 ** 9, n).astype(np.uint64)
    qs = rng.integers(0, 10 ** 9, n).astype(np.uint64)
    batch = eng.make_batch(ops, op_off, np.asarray(strands, dtype=np.uint8))
    d_ts, d_qs = eng.upload(ts), eng.upload(qs)
    cnt = eng.cigar_dotplot(batch, cutoff, d_ts, d_qs)
    off = eng.exclusive_scan_u64(n, cnt)
    oo, cc = off.numpy(), cnt.numpy()
    segs = eng.empty((int(oo[-1]) + 1) * 5, np.uint64).fill(0x2323232323232323)
    eng.cigar_dotplot(batch, cutoff, d_ts, d_qs, segs=segs, seg_off=off)
    sg = segs.numpy().reshape(-1, 5)
    def text_merged(sl):
        toks = []
        for w in sl.tolist():
            c, ln = w & 15, w >> 4
            if c in (9, 10) and toks:
                toks[-1][0] += ln
            else:
                toks.append([ln, synth.OP_CHARS[c] if c < 9 else "B"])
        return "".join("%d%s" % (ln, ch) for ln, ch in toks)
    for i in range(n):
        sl = ops[int(op_off[i]):int(op_off[i + 1])]
        if len(sl) == 0:
            assert int(cc[i]) == 0
            continue
        want = orc.cigar_to_base_plotdata("cg:Z:" + text_merged(sl), int(ts[i]), int(qs[i]), strands[i], cutoff)
        got = sg[int(oo[i]):int(oo[i + 1])]
        assert int(cc[i]) == len(want), (i, int(cc[i]), len(want))
        assert (got == want).all(), (i, got[:4], want[:4])
    assert (sg[int(oo[-1]):] == 0x2323232323232323).all()


def check_dotplot_long_records(eng, mops=2):
    """K12 with records cut into pieces: small knobs on nasty cuts (an M segment open across many pieces, breaks right at a
    cut, pieces without any event), a split indel in a long record (the serial walk), then one record of `mops` million ops"""
    eng.set_param("op_long_ops", 600)
    eng.set_param("op_piece_ops", 256)
    try:
        for seed, cutoff in ((5, 0), (5, 50), (6, 5)):
            ops, off = long_record_ops(seed, 3, 2300)
            ops = ops.copy()
            ops[(ops & 15) == 9] = (7 << 4) | 7          # no continuation pieces here: the piece walk itself
            ops[(ops >> 4) == (1 << 28) - 1] = (9 << 4) | 1
            n = len(off) - 1
            check_dotplot(eng, ops, off, [k & 1 for k in range(n)], cutoff)
        # a long record of small indels only behind its first op: one M segment over every piece; pieces of I ops only
        mk = lambda *p: [(ln << 4) | c for c, ln in p]
        rec = mk((7, 9)) + mk((1, 2), (2, 1)) * 700 + mk((7, 4)) + mk((1, 1)) * 600 + mk((2, 80), (7, 3))
        ops = np.array(rec, dtype=np.uint32)
        off = np.array([0, len(rec)], dtype=np.uint64)
        for cutoff in (0, 5, 100):
            check_dotplot(eng, ops, off, [1], cutoff)
        ops, off = long_record_ops(8, 2, 1500)           # split indels: continuation pieces -> serial walk
        check_dotplot(eng, ops, off, [0] * (len(off) - 1), 10)
        ops, off = long_record_ops(9, 2, 2300, shorts=12)  # mostly short records: two kernels share the batch
        check_dotplot(eng, ops, off, [k & 1 for k in range(len(off) - 1)], 3)
    finally:
        eng.set_param("op_long_ops", 16384)
        eng.set_param("op_piece_ops", 8192)
    if mops:
        ops, off = long_record_ops(7, 1, mops * 1_000_000)
        ops = ops.copy()
        ops[(ops & 15) == 9] = (7 << 4) | 7
        ops[(ops >> 4) == (1 << 28) - 1] = (9 << 4) | 1
        check_dotplot(eng, ops, off, [1] * (len(off) - 1), 30)


def check_dotplot_maf(eng, pairs, strands, cutoff):
    """MAF rows -> K3 runs -> ops (K11) -> segments == parse_maf_to_base_plotdata (cigar.rs:955-985)"""
    n = len(pairs)
    buf, t_off, q_off, cols = bytearray(b"@@@"), [], [], []
    for t, q in pairs:
        t_off.append(len(buf))
        buf += t + b"@"
        q_off.append(len(buf))
        buf += q + b"@@"
        cols.append(min(len(t), len(q)))
    rows = eng.upload(np.frombuffer(bytes(buf), dtype=np.uint8))
    d_t, d_q = eng.upload(np.array(t_off, dtype=np.uint64)), eng.upload(np.array(q_off, dtype=np.uint64))
    d_c, d_s = eng.upload(np.array(cols, dtype=np.uint64)), eng.upload(np.array(strands, dtype=np.uint8))
    counts, run_cnt = eng.maf_pair_stat(n, rows, d_t, d_q, d_c, d_s)
    run_off = eng.exclusive_scan_u64(n, run_cnt)
    ne = int(run_off.numpy()[-1])
    runs = eng.empty(ne + 1, np.uint64).fill(0)
    eng.maf_pair_stat(n, rows, d_t, d_q, d_c, d_s, counts=counts, run_cnt=run_cnt, runs=runs, run_off=run_off)
    ocnt = eng.maf_runs_ops(n, ne, runs, run_off, d_c)
    ooff = eng.exclusive_scan_u64(n, ocnt)
    nops = int(ooff.numpy()[-1])
    ops = eng.empty(nops + 4, np.uint32).fill(0)
    eng.maf_runs_ops(n, ne, runs, run_off, d_c, out=ops, out_off=ooff)
    batch = eng.make_batch_device(ops, ooff, d_s, n, nops)
    ts = np.arange(n, dtype=np.uint64) * np.uint64(1000) + np.uint64(7)
    qs = np.arange(n, dtype=np.uint64) * np.uint64(3000) + np.uint64(11)
    d_ts, d_qs = eng.upload(ts), eng.upload(qs)
    cnt = eng.cigar_dotplot(batch, cutoff, d_ts, d_qs)
    off = eng.exclusive_scan_u64(n, cnt)
    oo = off.numpy()
    segs = eng.empty((int(oo[-1]) + 1) * 5, np.uint64).fill(0)
    eng.cigar_dotplot(batch, cutoff, d_ts, d_qs, segs=segs, seg_off=off)
    sg = segs.numpy().reshape(-1, 5)
    for i, (t, q) in enumerate(pairs):
        want = orc.maf_to_base_plotdata(t, q, int(ts[i]), int(qs[i]), strands[i], cutoff)
        got = sg[int(oo[i]):int(oo[i + 1])]
        assert len(got) == len(want) and (got == want).all(), (i, got[:4], want[:4])


# ------------------------------------------------------------------------------------------------
# K13 PAF field splitter (+ the tokeniser on spans)
# ------------------------------------------------------------------------------------------------
def expected_paf_lines(text):
    """plain-Python restatement of what the csv reader + PafRecord deserialiser (paf.rs:24-30,50-78) do with a
    line that needs no csv state machine; everything else is FALLBACK"""
    def u64(b):
        b2 = b[1:] if b[:1] == b"+" else b
        if not b2 or not b2.isdigit() or not all(48 <= c <= 57 for c in b2) or int(b2) > 0xFFFFFFFFFFFFFFFF:
            return None
        return int(b2)
    out, pos = [], 0
    if not text:
        return out
    lines = text.split(b"\n")
    if text.endswith(b"\n"):
        lines.pop()
    for ln in lines:
        start = pos
        pos += len(ln) + 1
        if ln == b"" or ln[:1] == b"#":
            out.append(dict(status=2 if b"\r" in ln else 1))
            continue
        if b'"' in ln or b"\r" in ln:
            out.append(dict(status=2))
            continue
        f = ln.split(b"\t")
        offs = np.cumsum([0] + [len(x) + 1 for x in f])[:-1] + start
        nums = [u64(f[k]) if k < len(f) else None for k in (1, 2, 3, 6, 7, 8, 9, 10, 11)]
        if len(f) < 12 or any(v is None for v in nums) or f[4] not in (b"+", b"-"):
            out.append(dict(status=2))
            continue
        cg = next((k for k in range(12, len(f)) if f[k][:5] == b"cg:Z:"), None)
        cs = any(x[:5] == b"cs:Z:" for x in f[12:])
        if cg is None and cs:
            out.append(dict(status=2))
            continue
        out.append(dict(status=0, num=nums, neg=f[4] == b"-", qname=(int(offs[0]), len(f[0])), tname=(int(offs[5]), len(f[5])),
                        cg=None if cg is None else (int(offs[cg]) + 5, int(offs[cg]) + len(f[cg])), n_fields=len(f)))
    return out


def check_paf_split(eng, text):
    text = bytes(text)
    d_text = eng.upload(np.frombuffer(text + b"\0" * 16, dtype=np.uint8))
    n = eng.paf_split(d_text, len(text))
    want = expected_paf_lines(text)
    assert n == len(want), (n, len(want))
    if n == 0:
        return
    lines = eng.empty(n + 1, engine.PAF_LINE_DTYPE).fill(0xEE)
    assert eng.paf_split(d_text, len(text), lines) == n
    got = lines.numpy()
    spans = []
    for j, w in enumerate(want):
        g = got[j]
        assert int(g["status"]) == w["status"], (j, int(g["status"]), w)
        if w["status"]:
            continue
        assert [int(x) for x in g["num"]] == w["num"], (j, g["num"], w["num"])
        assert bool(g["strand_neg"]) == w["neg"] and int(g["n_fields"]) == w["n_fields"], j
        assert (int(g["qname_off"]), int(g["qname_len"])) == w["qname"], j
        assert (int(g["tname_off"]), int(g["tname_len"])) == w["tname"], j
        if w["cg"] is None:
            assert int(g["cg_beg"]) == NONE, j
        else:
            assert (int(g["cg_beg"]), int(g["cg_end"])) == w["cg"], (j, g["cg_beg"], g["cg_end"], w["cg"])
            spans.append(w["cg"])
    assert (got[n:n + 1].view(np.uint8) == 0xEE).all()
    if spans:       # the span tokeniser == the CSR tokeniser on the extracted texts
        m = len(spans)
        beg = eng.upload(np.array([a for a, _ in spans], dtype=np.uint64))
        end = eng.upload(np.array([b for _, b in spans], dtype=np.uint64))
        cnt, err = eng.cigar_tokenise_spans(m, d_text, beg, end)
        blob = b"".join(text[a:b] for a, b in spans)
        toff = np.cumsum([0] + [b - a for a, b in spans]).astype(np.uint64)
        d_blob = eng.upload(np.frombuffer(blob + b"0" * 64, dtype=np.uint8))
        cnt2, err2 = eng.cigar_tokenise(m, d_blob, eng.upload(toff))
        assert (cnt.numpy() == cnt2.numpy()).all() and (err.numpy() == err2.numpy()).all()
        off = eng.exclusive_scan_u64(m, cnt)
        tot = int(off.numpy()[-1])
        o1 = eng.empty(tot + 4, np.uint32).fill(0)
        o2 = eng.empty(tot + 4, np.uint32).fill(0)
        eng.cigar_tokenise_spans(m, d_text, beg, end, op_cnt=cnt, err=err, ops=o1, op_off=off)
        eng.cigar_tokenise(m, d_blob, eng.upload(toff), op_cnt=cnt2, err=err2, ops=o2, op_off=off)
        ok = np.array([e == 0 for e in err.numpy()["err"]])
        a1, a2, oo = o1.numpy(), o2.numpy(), off.numpy()
        for k in range(m):
            if ok[k]:
                assert (a1[int(oo[k]):int(oo[k + 1])] == a2[int(oo[k]):int(oo[k + 1])]).all(), k


# ------------------------------------------------------------------------------------------------
# K14 MAF line splitter
# ------------------------------------------------------------------------------------------------
def expected_maf_lines(text):
    """what MAFReader + parse_sline (maf.rs:25-36,138-211,371-421) make of every line that needs no
    Unicode-aware splitting; status 0 s-line, 1 other, 2 fallback"""
    def u64(b):
        b2 = b[1:] if b[:1] == b"+" else b
        if not b2 or not all(48 <= c <= 57 for c in b2) or int(b2) > 0xFFFFFFFFFFFFFFFF:
            return None
        return int(b2)
    out, pos = [], 0
    if not text:
        return out
    lines = text.split(b"\n")
    if text.endswith(b"\n"):
        lines.pop()
    for j, ln in enumerate(lines):
        start = pos
        pos += len(ln) + 1
        if j == 0 or ln[:1] != b"s":
            out.append(dict(status=1))
            continue
        toks, k = [], 0          # (offset, bytes) of the white-space separated tokens
        while k < len(ln):
            if ln[k] in b" \t\r\x0b\x0c":
                k += 1
                continue
            k2 = k
            while k2 < len(ln) and ln[k2] not in b" \t\r\x0b\x0c":
                k2 += 1
            toks.append((start + k, ln[k:k2]))
            k = k2
        if any(c >= 0x80 for c in ln):
            out.append(dict(status=2))
            continue
        nums = [u64(toks[i][1]) if i < len(toks) else None for i in (2, 3, 5)]
        if len(toks) != 7 or any(v is None for v in nums) or toks[4][1] not in (b"+", b"-"):
            out.append(dict(status=2))
            continue
        out.append(dict(status=0, num=nums, neg=toks[4][1] == b"-", name=(toks[1][0], len(toks[1][1])),
                        seq=(toks[6][0], len(toks[6][1]))))
    return out


def check_maf_split(eng, text):
    text = bytes(text)
    d_text = eng.upload(np.frombuffer(text + b"\0" * 16, dtype=np.uint8))
    n = eng.maf_split(d_text, len(text))
    want = expected_maf_lines(text)
    assert n == len(want), (n, len(want))
    if n == 0:
        return
    lines = eng.empty(n + 1, engine.MAF_LINE_DTYPE).fill(0xEE)
    assert eng.maf_split(d_text, len(text), lines) == n
    got = lines.numpy()
    for j, w in enumerate(want):
        g = got[j]
        assert int(g["status"]) == w["status"], (j, int(g["status"]), w)
        if w["status"]:
            continue
        assert [int(x) for x in g["num"]] == w["num"], (j, g["num"], w["num"])
        assert bool(g["strand_neg"]) == w["neg"], j
        assert (int(g["name_off"]), int(g["name_len"])) == w["name"], j
        assert (int(g["seq_off"]), int(g["seq_len"])) == w["seq"], j
    assert (got[n:n + 1].view(np.uint8) == 0xEE).all()


# ------------------------------------------------------------------------------------------------
# K15 FASTA text -> sequence pool (against the host faidx reader's semantics, restated here)
# ------------------------------------------------------------------------------------------------
def host_fasta_pool(text):
    """what the host reader builds (wga_host.cpp Faidx::load, htslib faidx semantics for well-formed files): the bytes
    of the lines behind the first header minus line ends (one '\\r' in front of a '\\n' / at the end of the file goes
    too), contigs in file order; name = up to the first white space"""
    pool, contigs, cur = bytearray(), [], None
    for line in text.split(b"\n"):
        if line.endswith(b"\r"):
            line = line[:-1]
        if line.startswith(b">"):
            if cur is not None:
                contigs.append((cur[0], cur[1], len(pool) - cur[1]))
            nm = line[1:].split(None, 1)[0] if line[1:].split(None, 1) else b""
            if line[1:2].isspace():
                nm = b""
            cur = (nm, len(pool))
        elif cur is not None:
            pool += line
    if cur is not None:
        contigs.append((cur[0], cur[1], len(pool) - cur[1]))
    return bytes(pool), contigs


FASTA_CASES = [
    b"", b"ACGT\n", b">a\n", b">a", b">a\nACGT", b">a\nACGT\n", b"junk before\nmore\n>c1 desc here\nACGT\nAC\n>c2\n\nGG\n",
    b">x\r\nAC\r\nGT\r\n>y\r\nTT\r", b">e1\n>e2\n>e3\nA\n>e4\n", b">t\tname\nNNNNacgtRYKM\n\n\nAC\n",
    b">s\nAC>GT\n>\nTT\n> spaced\nGG\n", b">dup\nAAAA\n>dup\nCCCC\n",
]


def check_fasta_pool(eng, text):
    blob = np.frombuffer(text + b"\0", dtype=np.uint8)
    d_text = eng.upload(blob)
    pool, contigs = eng.fasta_pool(d_text, len(text))
    want_pool, want = host_fasta_pool(text)
    got_pool = pool.numpy().tobytes()[: len(want_pool)] if len(want_pool) else b""
    assert pool.shape[0] == len(want_pool), (pool.shape, len(want_pool), text[:60])
    assert got_pool == want_pool, text[:60]
    assert len(contigs) == len(want), (len(contigs), len(want), text[:60])
    for c, (nm, off, ln) in zip(contigs, want):
        hs = int(c["hdr_start"])
        assert text[hs:hs + 1] == b">"
        line = text[hs + 1:int(c["hdr_end"])]
        got_nm = b"" if (not line or line[:1].isspace()) else line.split(None, 1)[0]
        assert got_nm == nm and int(c["pool_off"]) == off and int(c["len"]) == ln, (c, nm, off, ln, text[:60])


def random_fasta(rng, n_contigs, max_len, width=None, crlf=False):
    eol = b"\r\n" if crlf else b"\n"
    out = bytearray()
    for k in range(n_contigs):
        out += b">ctg%d some description" % k + eol
        seq = rand_seq(rng, int(rng.integers(0, max_len)), b"ACGTacgtNn")
        w = width or int(rng.integers(1, 120))
        for a in range(0, len(seq), w):
            out += seq[a:a + w] + eol
        if rng.random() < 0.2:
            out += eol                       # a blank line
    if out and rng.random() < 0.5:
        out = out[:-len(eol)]                # no line end at the end of the file
    return bytes(out)


# ------------------------------------------------------------------------------------------------
# K17 BGZF inflate (RFC 1951 on the device) against zlib
# ------------------------------------------------------------------------------------------------
def bgzf_member(data, level=6, strategy=0):
    """one BGZF member (RFC 1952 header with the BC extra field, raw DEFLATE, CRC32, ISIZE) of at most 64 KiB of data"""
    import struct, zlib
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    payload = co.compress(data) + co.flush()
    bsize = 12 + 6 + len(payload) + 8
    assert bsize <= 65536
    return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + payload
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def bgzf_table(img):
    """the member table of a BGZF image: (in_off, in_len, out_len, out_off) per member"""
    import struct
    rows, p, out = [], 0, 0
    while p < len(img):
        xlen = struct.unpack_from("<H", img, p + 10)[0]
        bsize = struct.unpack_from("<H", img, p + 16)[0] + 1
        isize = struct.unpack_from("<I", img, p + bsize - 4)[0]
        rows.append((p + 12 + xlen, bsize - 12 - xlen - 8, isize, out))
        out += isize
        p += bsize
    return np.array(rows, dtype=[("in_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4"), ("out_off", "<u8")]), out


def check_bgzf_inflate(eng):
    """every kind of DEFLATE block through wga_bgzf_inflate: stored (level 0), fixed codes (Z_FIXED), dynamic codes at several
    levels; literals only, long matches at distance 1 (run-length) and near 32 KiB, 258-byte matches, an empty member (the EOF
    marker), members of one byte and of 65 280 bytes; damaged streams are reported and write nothing outside their range"""
    import zlib
    rng = np.random.default_rng(17)
    dna = bytes(rng.choice(list(b"ACGTacgtN\n"), 60000, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02]).astype(np.uint8))
    rep = (b"ACGTTGCA" * 4000)[:30000] + dna[:2000] + b"A" * 20000 + dna[:2000]
    far = dna[:200] + bytes(rng.integers(0, 256, 32500, dtype=np.uint8)) + dna[:200] * 3
    datas = [(dna, 6, 0), (dna, 1, 0), (dna, 9, 0), (dna[:65280 - 9000], 0, 0), (rep, 6, 0), (rep, 6, zlib.Z_FIXED), (far, 9, 0),
             (b"", 6, 0), (b"x", 6, 0), (b"ab" * 129 + b"c", 6, zlib.Z_FIXED), (bytes(rng.integers(0, 256, 50000, dtype=np.uint8)), 6, 0),
             (b"N" * 65280, 6, 0), (dna[:777], 6, zlib.Z_HUFFMAN_ONLY), (rep[:40000], 6, zlib.Z_RLE)]
    img = b"".join(bgzf_member(d, lv, st) for d, lv, st in datas)
    want = b"".join(d for d, _, _ in datas)
    tab, total = bgzf_table(img)
    assert total == len(want)
    d_in = eng.upload(np.frombuffer(img + b"\0" * 16, dtype=np.uint8))
    d_tab = eng.upload(tab.view(np.uint8))
    out = eng.empty(total + 64, np.uint8).fill(0x23)
    status = eng.empty(len(tab), np.uint32).fill(0xFF)
    eng.bgzf_inflate(d_in, len(img), len(tab), d_tab, out, status)
    st = status.numpy()
    assert (st == 0).all(), st
    got = out.numpy()
    assert got[:total].tobytes() == want and (got[total:] == 0x23).all()
    # damaged streams: a flipped bit in the middle of every non-trivial member, a truncated payload, a wrong ISIZE
    bad = bytearray(img)
    for k in (0, 4, 6):
        at = int(tab["in_off"][k]) + int(tab["in_len"][k]) // 2
        bad[at] ^= 0x10
    tab2 = tab.copy()
    tab2["in_len"][1] = tab2["in_len"][1] // 2          # the input ends inside the stream
    tab2["out_len"][2] = tab2["out_len"][2] - 5          # fewer bytes than the stream holds
    out.fill(0x23)
    status.fill(0xFF)
    eng.bgzf_inflate(eng.upload(np.frombuffer(bytes(bad) + b"\0" * 16, dtype=np.uint8)), len(img), len(tab), eng.upload(tab2.view(np.uint8)),
                     out, status)
    st = status.numpy()
    assert st[1] != 0 and st[2] != 0, st
    for k in (0, 4, 6):                                   # a flipped bit: an error, or (rarely) other bytes of the same length
        a, n = int(tab["out_off"][k]), int(tab["out_len"][k])
        assert st[k] != 0 or out.numpy()[a:a + n].tobytes() != want[a:a + n], k
    got = out.numpy()
    assert (got[total:] == 0x23).all()
    for k in range(len(tab)):                             # untouched members are still right
        if k not in (0, 1, 2, 4, 6):
            a, n = int(tab["out_off"][k]), int(tab["out_len"][k])
            assert st[k] == 0 and got[a:a + n].tobytes() == want[a:a + n], k


# ------------------------------------------------------------------------------------------------
# K18 BGZF deflate on the device: zlib is the judge (the stream must inflate to the input), the member
# structure is read by bgzf_table, and the device inflate (K17) must take it back as well
# ------------------------------------------------------------------------------------------------
def bgzf_check_stream(img, want, eof_marker, one_call=True):
    """img is a valid BGZF stream of `want`: gzip inflates it to the input; every member carries the BC field with its own size,
    its CRC-32 and ISIZE, holds at most 32 768 input bytes and inflates alone; the 28-byte marker closes it when asked for"""
    import gzip, struct, zlib
    assert gzip.decompress(img) == want if img else want == b""
    p, got, sizes = 0, 0, []
    while p < len(img):
        assert img[p:p + 4] == b"\x1f\x8b\x08\x04" and img[p + 10:p + 16] == b"\x06\x00BC\x02\x00", p
        bsize = struct.unpack_from("<H", img, p + 16)[0] + 1
        crc, isize = struct.unpack_from("<II", img, p + bsize - 8)
        raw = zlib.decompressobj(-15)
        data = raw.decompress(img[p + 18:p + bsize - 8])
        assert raw.eof and raw.unused_data == b"" and len(data) == isize and isize <= (32768 if one_call else 65280)   # the host's zlib members hold up to 0xff00
        assert data == want[got:got + isize] and zlib.crc32(data) & 0xFFFFFFFF == crc
        sizes.append(isize)
        got += isize
        p += bsize
    assert p == len(img) and got == len(want)
    if eof_marker:
        assert sizes and sizes[-1] == 0 and img[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
        sizes = sizes[:-1]
    assert all(s > 0 for s in sizes) and (not one_call or all(s == 32768 for s in sizes[:-1]))   # one call: full members but the last
    return len(sizes)


def bgzf_deflate_inputs():
    rng = np.random.default_rng(18)
    dna = bytes(rng.choice(list(b"ACGT-acgtN\n"), 200000, p=[.2, .2, .2, .2, .1, .02, .02, .02, .02, .01, .01]).astype(np.uint8))
    maf = (b"##maf version=1\n" + b"".join(b"a score=0\ns chr%d %d 700 + 100000 " % (k, 13 * k) + dna[700 * k:700 * k + 700] +
                                          b"\ns qry%d 5 700 - 9000 " % k + dna[700 * k + 350:700 * k + 1050] + b"\n\n" for k in range(120)))
    skew = bytes(rng.choice(256, 70000, p=np.r_[0.97, np.full(255, 0.03 / 255)]).astype(np.uint8))   # one symbol nearly alone: 1-bit code
    fib = [1, 1]
    while sum(fib) + fib[-1] + fib[-2] <= 32768:
        fib.append(fib[-1] + fib[-2])
    deep = b"".join(bytes([65 + i]) * c for i, c in enumerate(fib))            # Fibonacci counts: a tree deeper than 15 before the halving
    return [("maf", maf), ("dna", dna), ("empty", b""), ("one byte", b"x"), ("two symbols", b"ab" * 5000),
            ("one symbol", b"N" * 40000), ("a member exactly", dna[:32768]), ("two members exactly", dna[:65536]),
            ("one over", dna[:32769]), ("random bytes (stored)", bytes(rng.integers(0, 256, 50000, dtype=np.uint8))),
            ("all 256 symbols, skewed", skew), ("deep tree", deep + dna[:1000]), ("127 bytes", dna[:127]), ("129 bytes", dna[:129])]


def check_bgzf_deflate(eng, inflate_too=True):
    """wga_bgzf_compress on every kind of input (see bgzf_deflate_inputs) at every alignment of input and output; pieces
    compressed by separate calls concatenate into one stream; a buffer that is too small is refused with the size it needs"""
    from wgatools_amd import _lib
    for k, (name, data) in enumerate(bgzf_deflate_inputs()):
        ia, oa = k % 4, (k * 3 + 1) % 4
        d_in = eng.upload(np.frombuffer(b"\xAA" * ia + data + b"\xBB" * 8, dtype=np.uint8))
        cap = int(eng.lib.wga_bgzf_bound(len(data)))
        out = eng.empty(oa + cap + 8, np.uint8).fill(0x23)
        marker = k % 3 != 1
        _, used = eng.bgzf_compress(d_in, len(data), out=out, eof_marker=marker, in_offset=ia, out_offset=oa)
        got = out.numpy()
        assert used <= cap and (got[:oa] == 0x23).all() and (got[oa + used:] == 0x23).all(), name
        img = got[oa:oa + used].tobytes()
        n_members = bgzf_check_stream(img, data, marker)
        assert n_members == (len(data) + 32767) // 32768, name
        if name in ("maf", "dna", "one symbol", "all 256 symbols, skewed"):
            assert used < len(data) * 0.45, (name, used, len(data))        # literals under their own code: four bases -> about 2.3 bits
        if name == "random bytes (stored)":
            assert used == len(data) + n_members * 31 + (28 if marker else 0)
        if inflate_too and data:
            tab, total = bgzf_table(img)
            back = eng.empty(total + 16, np.uint8).fill(0x23)
            status = eng.empty(len(tab), np.uint32).fill(0xFF)
            eng.bgzf_inflate(eng.upload(np.frombuffer(img + b"\0" * 16, dtype=np.uint8)), len(img), len(tab), eng.upload(tab.view(np.uint8)), back, status)
            assert (status.numpy() == 0).all() and back.numpy()[:total].tobytes() == data, name
    # pieces: three calls into one buffer make one stream; the last one closes it
    data = bgzf_deflate_inputs()[0][1]
    cuts = [0, 40001, 40001 + 32768, len(data)]
    d_in = eng.upload(np.frombuffer(data, dtype=np.uint8))
    out = eng.empty(int(eng.lib.wga_bgzf_bound(len(data))) + 200, np.uint8).fill(0)
    at = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        _, used = eng.bgzf_compress(d_in, b - a, out=out, eof_marker=b == len(data), in_offset=a, out_offset=at)
        at += used
    import gzip
    assert gzip.decompress(out.numpy()[:at].tobytes()) == data
    # too small a buffer: refused, the size reported, nothing written
    import ctypes as C
    small = eng.empty(1000, np.uint8).fill(0x23)
    need = C.c_uint64(0)
    rc = eng.lib.wga_bgzf_compress(eng.ctx, d_in.ptr, len(data), small.ptr, 1000, C.byref(need), 1)
    _, whole = eng.bgzf_compress(d_in, len(data))
    exact = C.c_uint64(0)                                      # the count call: the exact size, nothing written
    assert eng.lib.wga_bgzf_compress(eng.ctx, d_in.ptr, len(data), None, 0, C.byref(exact), 1) == 0 and exact.value == whole
    assert eng.lib.wga_bgzf_compress(eng.ctx, None, 0, None, 0, C.byref(exact), 1) == 0 and exact.value == 28
    assert eng.lib.wga_bgzf_compress(eng.ctx, None, 0, None, 0, C.byref(exact), 0) == 0 and exact.value == 0
    assert rc != 0 and need.value == whole and (small.numpy() == 0x23).all()
    assert b"wga_bgzf_bound" in eng.lib.wga_last_error()
