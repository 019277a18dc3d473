"""Parity of the product path — libwgahip.so on a real MI355X, called through the C-ABI —
against the CPU oracle.  `pytest -m gpu`.  Bit-exact everywhere."""
import os

import numpy as np
import pytest

import oracle_py as orc
import parity_cases as pc
from wgatools_amd import engine, synth

pytestmark = pytest.mark.gpu


# ---- the emulator cases again, on hardware ----------------------------------------------------
@pytest.mark.parametrize("seed,n,mean,use_m", [(1, 12, 700, False), (2, 60, 40, False),
                                               (3, 400, 3, True), (4, 2, 6000, False),
                                               (5, 3000, 200, False)])
def test_stat_random(gpu, seed, n, mean, use_m):
    pc.check_stat(gpu, synth.make_paf_batch(seed, n, mean, 60000, use_m=use_m))


@pytest.mark.parametrize("seed,n,mean,pool,pre,use_m", [
    (1, 12, 700, 50000, False, False), (2, 40, 60, 20000, True, False),
    (3, 300, 3, 5000, True, True), (4, 3, 5000, 200000, False, True),
    (6, 500, 400, 2_000_000, True, False)])
def test_paf2maf_random(gpu, seed, n, mean, pool, pre, use_m):
    b = synth.make_paf_batch(seed, n, mean, pool, use_m=use_m)
    rng = np.random.default_rng(seed)
    p = (rng.integers(0, 40, n), rng.integers(0, 40, n), rng.integers(0, 5, n)) if pre else None
    pc.check_paf2maf(gpu, b, pre=p)


def test_paf2maf_edge_cases(gpu):
    b = pc.edge_case_batch(gpu)
    n = len(b["strand_neg"])
    rng = np.random.default_rng(5)
    pc.check_paf2maf(gpu, b)
    pc.check_paf2maf(gpu, b, pre=(rng.integers(0, 33, n), rng.integers(0, 33, n), rng.integers(0, 3, n)))
    pc.check_paf2maf(gpu, b, force_slow=1)


def test_paf2maf_errors(gpu):
    cigars = ["10=", "10=", "6M1I", "3=1D", "4=2N4="]
    strands = [1, 1, 0, 0, 0]
    t = [b"ACGTACGTAC", b"ACGTACGTAC", b"ACGT", b"ACGT", b"ACGTACGT"]
    q = [b"ACGTRCGYAC", b"ACGTACGTAC", b"ACGTACG", b"AC", b"ACGTACGT"]
    r = pc.check_paf2maf(gpu, pc.batch_from_texts(gpu, cigars, strands, t, q))
    d = r["diag"]
    assert int(d["bad_base_pos"][0]) == 2 and int(d["bad_base_pos"][1]) == int(engine.NONE)
    assert int(d["panic_op_idx"][2]) == 1 and int(d["panic_op_idx"][3]) == 1
    assert int(d["bad_op_idx"][4]) == 1


def test_paf2maf_force_slow(gpu):
    pc.check_paf2maf(gpu, synth.make_paf_batch(5, 10, 300, 20000), force_slow=1)


# ---- BASELINE configs[1] shape at a size the oracle finishes in seconds -------------------------
def test_paf2maf_config2_reduced(gpu):
    """2 000 records x mean 5 kop (config 2 is 100 000): stat of every record + rows of a sample"""
    b = synth.make_paf_batch(0x5747415F + 2, 2000, 5000, 50_000_000)
    rng = np.random.default_rng(1)
    sample = sorted(rng.choice(2000, 60, replace=False).tolist())
    pc.check_stat(gpu, b, sample=range(0, 2000, 7))
    pc.check_paf2maf(gpu, b, sample=sample)


def test_paf2maf_long_cigar_stress(gpu):
    """>= 200 kop records (config 5 stress): linear-time numpy expectation"""
    b = synth.make_paf_batch(77, 6, 300_000, 40_000_000, sigma=0.05)
    r = pc.run_paf2maf(gpu, b)
    assert (r["diag"]["panic_op_idx"] == engine.NONE).all() and (r["diag"]["bad_base_pos"] == engine.NONE).all()
    for i in range(6):
        t = b["t_pool"][int(b["t_src_off"][i]):int(b["t_src_off"][i] + b["t_src_len"][i])].tobytes()
        q = b["q_pool"][int(b["q_src_off"][i]):int(b["q_src_off"][i] + b["q_src_len"][i])].tobytes()
        et, eq = pc.fast_expected_rows(pc.rec_ops(b, i), t, q, b["strand_neg"][i])
        to, qo = int(r["t_row_off"][i]), int(r["q_row_off"][i])
        assert r["out"][to:to + len(et)].tobytes() == et, i
        assert r["out"][qo:qo + len(eq)].tobytes() == eq, i


# ---- full BASELINE configs[1] size: size-independent properties ---------------------------------
def test_paf2maf_config2_full_size_properties(gpu):
    """100 000 records x mean 5 kop, 2 x 50 Mb pools (15 GB of rows).  Checked on device:
       * a checksum of checksums: per-letter byte counts of all rows == per-letter counts of all
         source slices (complemented for '-' strand) + exactly sum(I)+sum(D) gap bytes;
       * stat totals == class sums of the generator;
       * 48 records spread over the batch are bit-identical to the oracle."""
    import torch
    from wgatools_amd import pipeline
    dev = torch.device("cuda", 0)
    tb = synth.make_paf_batch_torch(0x5747415F + 2, 100_000, 5000, 50_000_000, dev)
    job = pipeline.Paf2MafStatJob(gpu, tb)
    job.bind_stream()
    job.out.fill_(0)
    torch.cuda.synchronize()
    job.step()
    torch.cuda.synchronize()
    assert bool((job.diag == -1).all())
    assert int(job.rec_off[-1]) == job.out_bytes
    # stat totals
    c = job.counts
    neg = tb["strand_neg"].bool()
    assert bool((c[:, 0] + c[:, 1] == tb["mx"]).all())
    assert bool((c[:, 3] + c[:, 7] == tb["i"]).all()) and bool((c[:, 5] + c[:, 9] == tb["d"]).all())
    assert bool((c[:, 10] == neg.long()).all()) and bool((c[~neg][:, 6:10] == 0).all())
    # letter histogram of the output, in 1 GB chunks
    hist = torch.zeros(256, dtype=torch.int64, device=dev)
    for a in range(0, job.out_bytes, 1 << 30):
        hist += torch.bincount(job.out[a:min(a + (1 << 30), job.out_bytes)].int(), minlength=256)
    # expected: prefix letter counts of the pools -> per-record slice counts
    exp = torch.zeros(256, dtype=torch.int64, device=dev)
    comp = {ord(a): ord(b) for a, b in zip("ACGTNacgtn", "TGCANtgcan")}
    def add(pool, off, ln, sel, complement):
        for ch in b"ACGTNacgtn":
            pre = torch.zeros(pool.numel() + 1, dtype=torch.int64, device=dev)
            torch.cumsum((pool == ch).long(), 0, out=pre[1:])
            cnt = (pre[off + ln] - pre[off])[sel].sum()
            exp[comp[ch] if complement else ch] += cnt
    allr = torch.ones_like(neg)
    add(tb["t_pool"], tb["t_src_off"], tb["t_src_len"], allr, False)
    add(tb["q_pool"], tb["q_src_off"], tb["q_src_len"], ~neg, False)
    add(tb["q_pool"], tb["q_src_off"], tb["q_src_len"], neg, True)
    exp[ord("-")] = int(tb["i"].sum() + tb["d"].sum())
    assert bool((hist == exp).all()), (hist.nonzero().flatten().tolist(),
                                        (hist - exp)[hist != exp].tolist())
    # sample vs oracle
    for i in torch.linspace(0, tb["n"] - 1, 48).long().tolist():
        r = synth.torch_batch_record_to_numpy(tb, i)
        assert job.record_rows(i) == pc.oracle_rows(r, 0), i


def test_paf2maf_maf2paf_roundtrip(gpu):
    """5e6 ops / 7e7 columns: K1 + K2 and K3 + K11 agree on every op and counter"""
    assert pc.check_paf2maf_maf2paf_roundtrip(gpu, 9, 1200, 4000) > 4_500_000


def test_paf2maf_drain_min_settings(gpu):
    pc.check_drain_min_settings(gpu, synth.make_paf_batch(21, 200, 900, 2_000_000))


def test_caller_owned_output_buffer(gpu):
    """the rows go where the caller says: a buffer of its own gives the bytes of a library-side allocation, a buffer that is
    too small is refused by the driver class before any launch"""
    import torch
    from wgatools_amd import pipeline
    dev = torch.device("cuda", 0)
    tb = synth.make_paf_batch_torch(78, 600, 3000, 5_000_000, dev)
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    b = pipeline.Paf2MafStatJob(gpu, tb)
    b.out.fill_(0x23)
    b.step()
    arena = torch.full((b.out_bytes * 2,), 0x23, dtype=torch.uint8, device=dev)
    a = pipeline.Paf2MafStatJob(gpu, tb, out=arena)
    a.step()
    torch.cuda.synchronize()
    assert a.out.data_ptr() == arena.data_ptr()
    assert bool((a.out[:a.out_bytes] == b.out[:b.out_bytes]).all()) and bool((a.diag == -1).all())
    small = torch.empty(16, dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        pipeline.Paf2MafStatJob(gpu, tb, out=small)
    gpu.reset_stream()


def test_paf2maf_stream_kernel(gpu):
    """the streaming row kernel (expand_variant 3, the default of every batch but those of tiny records) against the oracle"""
    pc.window_kernel_cases(gpu, variant=3)
    pc.stream_kernel_cases(gpu)
    pc.check_paf2maf(gpu, synth.make_paf_batch(6, 500, 400, 2_000_000), variant=3)
    pc.check_paf2maf(gpu, synth.make_paf_batch(12, 1, 300_000, 1_500_000, sigma=0.01), variant=3)   # one record over ~300 tiles
    for seed in range(30, 40):
        n = int(np.random.default_rng(seed).integers(1, 300))
        mean = int(np.random.default_rng(seed + 1).integers(1, 3000))
        b = synth.make_paf_batch(seed, n, mean, 300000, use_m=bool(seed & 1))
        rng = np.random.default_rng(seed)
        pc.check_paf2maf(gpu, b, pre=(rng.integers(0, 130, n), rng.integers(0, 130, n), rng.integers(0, 5, n)), variant=3)


def test_paf2maf_v1_kernel(gpu):
    """the row kernels' battery on v1 (expand_variant 0): the kernel of the tiles the streaming kernel leaves"""
    pc.window_kernel_cases(gpu, variant=0)
    pc.check_paf2maf(gpu, synth.make_paf_batch(6, 500, 400, 2_000_000), variant=0)
    b = pc.dense_indel_batch(gpu)
    pc.check_paf2maf(gpu, b, variant=0)
    pc.check_paf2maf(gpu, b, variant=0)
    bad = synth.make_paf_batch(13, 2, 3000, 100_000)   # an invalid base in a '-' strand row whose slice overlaps another record's
    bad["strand_neg"][:] = 1
    qp = bad["q_pool"].copy()
    qp[int(bad["q_src_off"][1] + bad["q_src_len"][1] // 2)] = ord("R")
    bad["q_pool"] = qp
    pc.check_paf2maf(gpu, bad, variant=0)
    pc.check_paf2maf(gpu, synth.make_paf_batch(12, 1, 300_000, 1_500_000, sigma=0.01), variant=0)   # one record over ~300 tiles


def test_paf2maf_kernels_agree_at_size(gpu):
    """the two row kernels (v1, streaming) over whole BASELINE-sized batches (configs[1], 500-op records, 30-op
    records): every byte of the output text identical, on the device"""
    import torch
    from wgatools_amd import pipeline
    dev = torch.device("cuda", 0)
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    for rec, mean, pool in [(100_000, 5000, 50), (1_000_000, 500, 50), (3_000_000, 30, 20)]:
        tb = synth.make_paf_batch_torch(0x5747415F + 2, rec, mean, pool * 1_000_000, dev)
        outs = []
        for v in (0, 3):
            gpu.set_param("expand_variant", v)
            job = pipeline.Paf2MafStatJob(gpu, tb, with_text=True)
            job.out.fill_(0x23)
            job.bind_stream()
            job.step()
            torch.cuda.synchronize()
            assert bool((job.diag == -1).all()) and gpu.get_param("expand_variant_used") == v
            outs.append(job.out)
            del job
        assert bool(torch.equal(outs[0], outs[1])), (rec, mean)
        del outs, tb
        torch.cuda.empty_cache()
    gpu.set_param("expand_variant", pc.DEFAULT_EXPAND_VARIANT)
    gpu.reset_stream()


@pytest.mark.parametrize("variant", [0, 3])
def test_paf2maf_wide_tile_auto_slow_path(gpu, variant):
    """one tile wider than 2^31 columns (9 D ops of 2^28-1) takes the u64 fallback by itself (variant 3: through the
    list of such tiles that the streaming kernel leaves to v1's op-serial walk)"""
    import torch
    gpu.set_param("expand_variant", variant)
    dev = torch.device("cuda", 0)
    big = (1 << 28) - 1
    ops = np.array([(5 << 4) | 7] + [(big << 4) | 2] * 9 + [(7 << 4) | 7], dtype=np.uint32)
    t_len, q_len = 12 + 9 * big, 12
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    t_pool = lut[torch.randint(0, 4, (t_len,), device=dev, generator=g)]
    q_pool = lut[torch.randint(0, 4, (q_len,), device=dev, generator=g)]
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    batch = gpu.make_batch(ops, np.array([0, len(ops)], dtype=np.uint64), np.array([0], dtype=np.uint8))
    counts, diag, tws = gpu.cigar_stat(batch)
    z = gpu.upload(np.zeros(1, dtype=np.uint64))
    tl, ql = gpu.upload(np.array([t_len], dtype=np.uint64)), gpu.upload(np.array([q_len], dtype=np.uint64))
    tro, qro, reco = gpu.paf2maf_layout(1, counts, tl, ql)
    total = int(reco.numpy()[-1])
    assert total == 2 * t_len
    out = torch.full((total + 64,), 0x23, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    gpu.paf2maf_expand(batch, counts, tws, t_pool, t_len, z, tl, q_pool, q_len, z, ql, out, tro, qro, diag)
    gpu.sync()
    assert bool((out[:t_len] == t_pool).all())            # D consumes the target: row == slice
    qrow = out[t_len:2 * t_len]
    assert bool((qrow[:5] == q_pool[:5]).all()) and bool((qrow[-7:] == q_pool[5:]).all())
    assert bool((qrow[5:-7] == 45).all()) and bool((out[total:] == 0x23).all())
    c = counts.numpy()[0]
    assert int(c["del_bp"]) == 9 * big and int(c["del_ev"]) == 9 and int(c["match"]) == 12
    gpu.set_param("expand_variant", pc.DEFAULT_EXPAND_VARIANT)


# ---- the other consumers ---------------------------------------------------------------------------
def _cov_problem(seed, n, mean, nt, tmax):
    rng = np.random.default_rng(seed)
    b = pc.sprinkle_ops(rng, synth.make_paf_batch(seed, n, mean, 1000))
    tlen = rng.integers(tmax // 10, tmax, nt)
    tid = rng.integers(0, nt, n)
    tstart = (rng.random(n) * tlen[tid] * 1.02).astype(np.uint64)
    return b, tid, tstart, tlen


@pytest.mark.parametrize("seed,n,mean,nt,tmax,align", [(1, 30, 40, 3, 3000, 4), (2, 200, 5, 7, 3000, 1),
                                                        (3, 4, 3000, 2, 90000, 4), (4, 3000, 300, 16, 400000, 4)])
def test_pafcov(gpu, seed, n, mean, nt, tmax, align):
    b, tid, tstart, tlen = _cov_problem(seed, n, mean, nt, tmax)
    pc.check_pafcov(gpu, b, tid, tstart, tlen, align=align)
    pc.check_pafcov(gpu, b, tid, tstart, tlen, align=align, split=True)


@pytest.mark.parametrize("base", [0, 1])
@pytest.mark.parametrize("seed,n,mean", [(1, 20, 60), (2, 150, 4), (3, 3, 2600), (4, 400, 500)])
def test_pafpseudo(gpu, base, seed, n, mean):
    rng = np.random.default_rng(seed)
    b = pc.sprinkle_ops(rng, synth.make_paf_batch(seed, n, mean, 600000), codes=(3, 5, 6, 11))
    ops = b["ops"].copy()
    k = rng.integers(0, len(ops), max(1, len(ops) // 50))
    ops[k] = (ops[k] & ~np.uint32(15)) | np.uint32(4)
    b["ops"] = ops
    code, length = ops & 15, (ops >> 4).astype(np.uint64)
    v = np.where((code == 0) | (code == 7) | (code == 8) | (code == 1) | (code == 4), length, 0).astype(np.uint64)
    c = np.concatenate([np.zeros(1, np.uint64), np.cumsum(v, dtype=np.uint64)])
    b["q_src_len"] = c[b["op_off"][1:].astype(np.int64)] - c[b["op_off"][:-1].astype(np.int64)]
    b["q_src_off"] = (rng.random(n) * (len(b["q_pool"]) - b["q_src_len"].astype(np.float64))).astype(np.uint64)
    seg = synth.class_sums(code, ops >> 4, b["op_off"])
    skip = np.where(rng.random(n) < 0.4, rng.integers(0, 30, n), 0)
    skip = np.minimum(skip, (seg["mx"] + seg["d"]).astype(np.int64))
    pc.check_pafpseudo(gpu, b, base, skip=None)
    pc.check_pafpseudo(gpu, b, base, skip=skip)


def test_pafpseudo_stream_kernel(gpu):
    pc.pseudo_stream_cases(gpu)


def test_pafpseudo_symbol_runs(gpu):
    """symbol mode, per-granule walk over the ops that cover it: long X / D runs, rows that start inside a granule, dense
    single-column ops, one op of 1.2 M columns, trimmed heads up to 8191 columns"""
    cigars = ["100=50X9000D3=1X40D8000=33X7=", "5=1200000D5=", "1=1X" * 600 + "4=", "70D", "70X", "3=2I4=1X5=",
              "20=" + "35X2=" * 300, "9000=1D9000=1X100="]
    strands = [0, 1, 0, 0, 1, 0, 0, 1]
    b = pc.batch_from_texts(gpu, cigars, strands, [b"A"] * 8, [b"A"] * 8)
    pc.check_pafpseudo(gpu, b, 0)
    pc.check_pafpseudo(gpu, b, 0, skip=[0, 3, 17, 0, 5, 2, 33, 8191])


def test_pafpseudo_length_mismatch(gpu):
    cigars = ["5=2I3=", "5=2I3=", "4=3D4=", "10=", "8=2I", "8=1D", "3=2S1="]
    strands = [0, 1, 1, 0, 0, 1, 0]
    q = [b"ACGTACGTACGTTT", b"ACGTACGTACGTTT", b"ACGTAC", b"ACGTA", b"ACGTA", b"ACGTA", b"ACGTA"]
    pc.check_pafpseudo(gpu, pc.batch_from_texts(gpu, cigars, strands, [b"A"] * 7, q), 1)


def test_maf_pair_stat(gpu):
    rng = np.random.default_rng(4)
    pairs, strands = [], []
    b = synth.make_paf_batch(21, 40, 400, 300000)
    for i in range(40):
        pairs.append(pc.oracle_rows(b, i))
        strands.append(int(b["strand_neg"][i]))
    for L in (0, 1, 15, 16, 17, 63, 64, 65, 200, 1000, 1023, 1024, 1025, 1041, 2100, 100000):
        pairs.append((pc.rand_seq(rng, L, b"ACGTacgt--N"), pc.rand_seq(rng, L + int(rng.integers(0, 3)), b"ACGTacgt--N")))
        strands.append(L & 1)
    from helpers import GOLDEN, read_maf_blocks
    blk = read_maf_blocks(os.path.join(GOLDEN, "test.maf"))[0]
    pairs.append((blk[0]["seq"], blk[1]["seq"]))
    strands.append(0)
    pairs += pc.binary_row_pairs(rng)
    strands += [0, 1, 0, 1]
    pc.check_maf_pair(gpu, pairs, strands)
    pc.check_maf_call_runs(gpu, pairs)


def test_scan(gpu):
    rng = np.random.default_rng(3)
    for n in (0, 1, 5, 1024, 1025, 5000, 3_000_000):
        v = rng.integers(0, 1 << 40, n).astype(np.uint64)
        got = gpu.exclusive_scan_u64(n, gpu.upload(v) if n else None).numpy()
        assert (got == np.concatenate([np.zeros(1, np.uint64), np.cumsum(v, dtype=np.uint64)])).all()


def test_paf_call_events(gpu):
    b = synth.make_paf_batch(33, 300, 3000, 400000)
    for svlen, snp in ((0, True), (3, False), (50, True)):
        assert pc.check_paf_call_events(gpu, b["ops"], b["op_off"], svlen, snp) > 0
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
        pc.check_paf_call_events(gpu, ops, off, svlen, snp)


def test_paf_call_long_records_in_pieces(gpu):
    pc.check_paf_call_long_records(gpu, mops=2)


def test_dotplot_long_records_in_pieces(gpu):
    pc.check_dotplot_long_records(gpu, mops=2)


def test_piece_table_kept_or_rebuilt(gpu):
    pc.check_piece_table_rebuild(gpu)


def test_cigar_chain_long_records_in_pieces(gpu):
    pc.check_cigar_chain_long_records(gpu, mops=2)


def test_device_tokeniser(gpu):
    pc.check_tokeniser(gpu, pc.TOKENISER_EDGE_TEXTS)
    b = synth.make_paf_batch(41, 400, 3000, 300000)
    texts = [synth.cigar_text(pc.rec_ops(b, i)).encode() for i in range(400)]
    pc.check_tokeniser(gpu, texts + [b"3M", b""] + texts[:3])


def test_pafcov_ops_across_many_windows(gpu):
    pc.check_pafcov_long_ops(gpu)


def test_pafcov_random_shapes(gpu):
    pc.check_pafcov_random(gpu, 11, 12)
    pc.check_pafcov_random(gpu, 12, 60)


def test_pafcov_look_back(gpu):
    """K5's list pass: tile sums by look-back over records of 2 .. 70 tiles (tests/parity_cases.py)"""
    pc.check_pafcov_look_back(gpu)


def test_pafcov_format(gpu):
    rng = np.random.default_rng(3)
    pc.check_pafcov_format(gpu, b"chr1", [0, 1, 9, 10, 99, 100, 2147483647, 12345], 0)
    pc.check_pafcov_format(gpu, b"g01#1#chr1", rng.integers(0, 500, 1300), 95)
    pc.check_pafcov_format(gpu, b"t", rng.integers(0, 3, 40), 999_999_990)
    pc.check_pafcov_format(gpu, b"big", rng.integers(0, 70000, 30), 9_999_999_990)
    pc.check_pafcov_format(gpu, b"huge", [7, 8], 18_446_744_073_709_551_000)
    pc.check_pafcov_format(gpu, b"", [5], 41)
    pc.check_pafcov_format(gpu, b"a_target_name_longer_than_the_staging_buffer_takes_512_lines_of", rng.integers(0, 500, 1100), 7)
    pc.check_pafcov_format(gpu, b"c", rng.integers(0, 9, 512 * 3), 999_999_000)   # exactly three blocks, a digit roll-over inside
    pc.check_pafcov_format(gpu, b"chr12", rng.integers(0, 300, 2_000_001), 99_000_000)
    pc.check_pafcov_format(gpu, b"none", [], 0)


def test_cigar_chain(gpu):
    b = synth.make_paf_batch(57, 300, 3000, 400000)
    pc.check_cigar_chain(gpu, b["ops"], b["op_off"])
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
    pc.check_cigar_chain(gpu, ops, off)


@pytest.mark.gpu
def test_cigar_chain_steps(gpu):
    for seed in (1, 2):
        ops, off = pc.chain_stress_records(seed)
        pc.check_cigar_chain(gpu, ops, off)


# ---- the other BASELINE configs at (or near) their stated sizes: properties checked on the device ------------
def _synthetic_maf_rows(dev, n, L, seed):
    """config 3 shaped rows: n blocks x L columns, 1.2 % SNP, 0.15 % indel-open with geometric lengths; the rows of
    the first 200 000 blocks are repeated"""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n0 = min(n, 200_000)
    tot = n0 * L
    alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    t = alpha[torch.randint(0, 4, (tot,), device=dev, generator=g)]
    q = t.clone()
    snp = torch.rand(tot, device=dev, generator=g) < 0.012
    q[snp] = alpha[torch.randint(0, 4, (int(snp.sum()),), device=dev, generator=g)]
    opn = torch.rand(tot, device=dev, generator=g) < 0.0015
    ln = torch.zeros(tot, dtype=torch.int32, device=dev)
    ln[opn] = torch.empty(int(opn.sum()), device=dev).geometric_(1 / 3.0, generator=g).to(torch.int32)
    idx = torch.arange(tot, device=dev)
    last_start = torch.cummax(torch.where(opn, idx, torch.zeros_like(idx)), 0).values
    in_gap = (idx - last_start < ln[last_start]) & (last_start > 0)
    which = last_start % 3
    t[in_gap & (which == 0)] = 45
    q[in_gap & (which == 1)] = 45
    both = in_gap & (which == 2) & (idx % 7 == 0)            # a few '-','-' columns
    t[both] = 45
    q[both] = 45
    rep = n // n0
    return t.repeat(rep), q.repeat(rep), n0 * rep


def test_runs_bridge_synthetic(gpu):
    L = (1 << 28) - 1
    recs = [[(5, 0), (3, 1), (2, 3), (4, 2)], [], [(L, 0), (L + 1, 1), (2 * L + 7, 2), (1, 3), (3 * L, 3)],
            [(1, c) for c in (0, 1, 2, 3)] * 200, [(10 ** 9, 2)], [(123456789, 0)]]
    pc.check_runs_bridge_synthetic(gpu, recs)


def test_pafpseudo_fill_without_the_class_sums_call(gpu):
    """wga_pafpseudo_fill takes the sums wga_cigar_class_sums left for the same batch — or computes them when no such call
    stands in front (other arrays, or none at all)"""
    b = synth.make_paf_batch(7, 40, 300, 60000)
    code, length = b["ops"] & 15, (b["ops"] >> 4).astype(np.uint64)
    v = np.where((code == 0) | (code == 7) | (code == 8) | (code == 1) | (code == 4), length, 0).astype(np.uint64)
    c = np.concatenate([np.zeros(1, np.uint64), np.cumsum(v, dtype=np.uint64)])
    b["q_src_len"] = c[b["op_off"][1:].astype(np.int64)] - c[b["op_off"][:-1].astype(np.int64)]
    b["q_src_off"] = np.zeros(40, dtype=np.uint64)
    for base in (0, 1):
        pc.check_pafpseudo(gpu, b, base, sums_call=False)
        pc.check_pafpseudo(gpu, b, base)
        pc.check_pafpseudo(gpu, b, base, sums_call=False)


def test_elem_scan_reuse(gpu):
    pc.check_elem_scan_reuse(gpu)


def test_bridge_blocks(gpu):
    pc.check_bridge_blocks(gpu)


def test_chain_lines(gpu):
    rng = np.random.default_rng(77)
    L = (1 << 28) - 1
    recs = [[(10, 2, 0), (5, 0, 3), (7, 1, 1), (4, 0, 0)], [(9, 0, 0)], [],
            [(0, 0, 4), (3, 0, 0), (0, 2, 0), (6, 0, 0)],              # zero sizes: "0M" in the text, no op
            [(L + 5, 2 * L + 1, L), (1, 0, 0)],                         # lengths beyond one packed op
            [(int(rng.integers(1, 50)), int(rng.integers(0, 4)), int(rng.integers(0, 4))) for _ in range(700)] + [(3, 0, 0)]]
    strands = [0, 1, 0, 1, 0, 1]
    pc.check_chain_lines(gpu, recs, strands)
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
    pc.check_chain_lines(gpu, recs2, strands2, seqs=seqs)



def test_dotplot_segments(gpu):
    b = synth.make_paf_batch(91, 14, 700, 500000)
    pc.sprinkle_ops(np.random.default_rng(2), b, frac=0.03)
    for cutoff in (0, 3, 50, 10 ** 6):
        pc.check_dotplot(gpu, b["ops"], b["op_off"], b["strand_neg"], cutoff)
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
        pc.check_dotplot(gpu, ops, off, strands, cutoff)
    rng = np.random.default_rng(12)
    pairs, st = [], []
    for L2 in (0, 1, 17, 64, 300, 1025, 2100):
        pairs.append((pc.rand_seq(rng, L2, b"ACGTacgt--N"), pc.rand_seq(rng, L2 + int(rng.integers(0, 3)), b"ACGTacgt--N")))
        st.append(L2 & 1)
    for cutoff in (0, 1, 5):
        pc.check_dotplot_maf(gpu, pairs, st, cutoff)



def test_paf_split(gpu):
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
    pc.check_paf_split(gpu, clean + b"\n")
    pc.check_paf_split(gpu, clean)                       # no newline at the end
    pc.check_paf_split(gpu, b"")
    pc.check_paf_split(gpu, b"\n\n")
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
    pc.check_paf_split(gpu, b"\n".join(odd) + b"\n" + clean)
    # delimiters right at the 16-byte / 4096-byte block edges
    for pad in (4080, 4095, 4096, 4097, 8191):
        pc.check_paf_split(gpu, b"#" + b"x" * (pad - 1) + b"\n" + rows[0].encode() + b"\n" + b"\t" * 40 + b"\n")



def test_maf_pair_long_rows_fold_lane_counters(gpu):
    """rows beyond 4095 steps of 1024 columns: the 16-bit lane counters are folded into the wave totals on the way
    (the emulator build of the CPU suite folds every 3 steps instead); counts against the oracle"""
    rng = np.random.default_rng(21)
    L = 4095 * 1024 * 2 + 1500
    t = pc.rand_seq(rng, L, b"ACGTacgt--N")
    q = pc.rand_seq(rng, L, b"ACGTacgt--N")
    pairs, strands = [(t, q), (t[:70000], q[:70000]), (t[:4095 * 1024], q[:4095 * 1024])], [1, 0, 0]
    buf, t_off, q_off = bytearray(b"@@@"), [], []
    for a, b in pairs:
        t_off.append(len(buf))
        buf += a + b"@"
        q_off.append(len(buf))
        buf += b + b"@@"
    rows = gpu.upload(np.frombuffer(bytes(buf), dtype=np.uint8))
    n = len(pairs)
    exp = [orc.parse_maf_seq_to_cigar(a, b, strands[i]) for i, (a, b) in enumerate(pairs)]
    # once as ONE wave per row (the fold of the lane counters), once piece by piece (the default beyond 32 768 columns)
    for long_cols in (1 << 62, 32768):
        gpu.set_param("maf_long_cols", long_cols)
        try:
            counts, run_cnt = gpu.maf_pair_stat(n, rows, gpu.upload(np.array(t_off, dtype=np.uint64)),
                                                gpu.upload(np.array(q_off, dtype=np.uint64)),
                                                gpu.upload(np.array([len(a) for a, _ in pairs], dtype=np.uint64)),
                                                gpu.upload(np.array(strands, dtype=np.uint8)))
        finally:
            gpu.set_param("maf_long_cols", 32768)
        c, rc = counts.numpy(), run_cnt.numpy()
        for i in range(n):
            exp_counts, exp_txt = exp[i]
            assert tuple(int(x) for x in c[i]) == exp_counts, (long_cols, i, exp_counts, c[i])
            assert int(rc[i]) == sum(1 for ch in exp_txt if not ch.isdigit())


def test_maf_long_blocks_piecewise(gpu):
    rng = np.random.default_rng(77)
    pairs, strands = [], []
    for L in (5, 63, 64, 65, 130, 999, 1024, 2100, 5000, 70000):
        t = pc.rand_seq(rng, L, b"ACGTacgt--N")
        q = pc.rand_seq(rng, L + int(rng.integers(0, 3)), b"ACGTacgt--N")
        pairs.append((t, q))
        strands.append(L & 1)
    pairs.append((b"-" * 700 + b"ACGT" * 100, b"-" * 650 + b"A" * 50 + b"ACGA" * 100))
    strands.append(1)
    try:
        for long_cols, piece_cols in ((100, 64), (1, 1000), (500, 17), (64, 1024), (32768, 16384)):
            gpu.set_param("maf_long_cols", long_cols)
            gpu.set_param("maf_piece_cols", piece_cols)
            pc.check_maf_pair(gpu, pairs, strands)
            pc.check_maf_call_runs(gpu, pairs)
    finally:
        gpu.set_param("maf_long_cols", 32768)
        gpu.set_param("maf_piece_cols", 16384)


def test_maf_block_of_1e8_columns(gpu):
    """ONE two-row block of 10^8 columns tiles across the chip (6 104 pieces): counters and
    run count against the oracle, the full run lists of both walks against an independent computation on the device,
    and the rate of the two count passes (printed; profiles/r02_other_kernels.txt)"""
    import torch
    dev = torch.device("cuda", 0)
    L = 100_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    alpha = torch.tensor(list(b"ACGTACGTACGTACGTACGTACGTACGTACG-"), dtype=torch.uint8, device=dev)   # 1/32 gaps
    t = alpha[torch.randint(0, 32, (L,), device=dev, generator=g)]
    q = torch.where(torch.rand(L, device=dev, generator=g) < 0.9, t, alpha[torch.randint(0, 32, (L,), device=dev, generator=g)])
    rows = torch.cat([t, torch.full((64,), 64, dtype=torch.uint8, device=dev), q, torch.full((64,), 64, dtype=torch.uint8, device=dev)])
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    d_t, d_q = gpu.upload(np.array([0], dtype=np.uint64)), gpu.upload(np.array([L + 64], dtype=np.uint64))
    d_c, d_s = gpu.upload(np.array([L], dtype=np.uint64)), gpu.upload(np.array([0], dtype=np.uint8))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    counts, run_cnt = gpu.maf_pair_stat(1, rows, d_t, d_q, d_c, d_s)              # warm-up
    ev[0].record()
    counts, run_cnt = gpu.maf_pair_stat(1, rows, d_t, d_q, d_c, d_s)
    ev[1].record()
    crun_cnt = gpu.maf_call_runs(1, rows, d_t, d_q, d_c)
    ev[2].record()
    torch.cuda.synchronize()
    ms3, ms4 = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    print("\n1e8-column block: K3 count pass %.3f ms = %.0f GB/s, K4 count pass %.3f ms = %.0f GB/s" % (
        ms3, 2 * L / ms3 / 1e6, ms4, 2 * L / ms4 / 1e6))
    # independent expectation on the device
    tg, qg = t == 45, q == 45
    cls3 = torch.where(t == q, 0, torch.where(tg, 1, torch.where(qg, 2, 3))).to(torch.uint8)
    cls4 = torch.where(tg & qg, 4, torch.where(tg, 1, torch.where(qg, 2, torch.where(t == q, 0, 3)))).to(torch.uint8)
    for cls, caller in ((cls3, False), (cls4, True)):
        start = torch.ones(L, dtype=torch.bool, device=dev)
        start[1:] = cls[1:] != cls[:-1]
        idx = torch.nonzero(start).flatten()
        nrun = int(idx.numel())
        if not caller:
            assert int(run_cnt.numpy()[0]) == nrun
            run_off = gpu.exclusive_scan_u64(1, run_cnt)
            runs = torch.zeros(nrun + 1, dtype=torch.int64, device=dev)
            gpu.maf_pair_stat(1, rows, d_t, d_q, d_c, d_s, counts=counts, run_cnt=run_cnt, runs=runs, run_off=run_off)
            gpu.sync()
            want = (idx << 3) | cls[idx].to(torch.int64)
            assert bool((runs[:nrun] == want).all())
        else:
            assert int(crun_cnt.numpy()[0]) == nrun
            run_off = gpu.exclusive_scan_u64(1, crun_cnt)
            runs = torch.zeros(3 * nrun + 3, dtype=torch.int64, device=dev)
            gpu.maf_call_runs(1, rows, d_t, d_q, d_c, run_cnt=crun_cnt, runs=runs, run_off=run_off)
            gpu.sync()
            r3 = runs[:3 * nrun].view(nrun, 3)
            tb = torch.cumsum((~tg).to(torch.int64), 0) - (~tg).to(torch.int64)
            qb = torch.cumsum((~qg).to(torch.int64), 0) - (~qg).to(torch.int64)
            assert bool((r3[:, 0] == ((idx << 3) | cls[idx].to(torch.int64))).all())
            assert bool((r3[:, 1] == tb[idx]).all()) and bool((r3[:, 2] == qb[idx]).all())
        del start, idx, runs
    # the oracle on the whole block: counters and run count
    exp_counts, exp_txt = orc.parse_maf_seq_to_cigar(t.cpu().numpy().tobytes(), q.cpu().numpy().tobytes(), 0)
    assert tuple(int(x) for x in counts.numpy()[0]) == exp_counts
    assert int(run_cnt.numpy()[0]) == sum(1 for ch in exp_txt if not ch.isdigit())
    gpu.reset_stream()


def test_maf_split(gpu):
    rng = np.random.default_rng(8)
    def block(k, cols, extra=b""):
        t = pc.rand_seq(rng, cols, b"ACGTacgt-N")
        q = pc.rand_seq(rng, cols, b"ACGTacgt-N")
        return (b"a score=%d\n" % k + b"s ref.chr%d   %d %d + 1000000 " % (k, 7 * k, cols) + t + b"\n" +
                b"s\tqry.%d\t%d\t%d\t-\t+2000000\t" % (k, 11 * k, cols) + q + extra + b"\n\n")
    clean = b"##maf version=1 scoring=x\n# a comment\n" + b"".join(block(k, c) for k, c in enumerate((1, 15, 16, 17, 300, 4100, 9000)))
    pc.check_maf_split(gpu, clean)
    pc.check_maf_split(gpu, clean[:-2])                      # no newline at the end
    pc.check_maf_split(gpu, b"")
    pc.check_maf_split(gpu, b"s first line is the header even if it looks like an s-line\ns a 1 2 + 3 ACGT\n")
    odd = [b"s a 1 2 + 3", b"s a 1 2 + 3 ACGT extra", b"s a x 2 + 3 ACGT", b"s a 1 2 * 3 ACGT", b"s a 1 2 + 18446744073709551616 ACGT",
           b"sX a 1 2 + 3 ACGT", b" s a 1 2 + 3 ACGT", b"s a 1 2 + 3 ACGT \r", b"s\x0ba\x0c1 2 + 3 ACGT", b"s a\xc2\xa01 2 + 3 ACGT",
           b"i a N 0 C 0", b"e a 1 2 + 3 I", b"q a 99", b"", b"s", b"s a +1 2 - 3 AC-GT"]
    pc.check_maf_split(gpu, b"##maf\n" + b"\n".join(odd) + b"\n" + clean)
    pc.check_maf_split(gpu, (b"##maf\n" + b"\n".join(odd)).replace(b"\n", b"\r\n"))
    for pad in (4079, 4095, 4096, 4097):                     # delimiters at the block edges
        pc.check_maf_split(gpu, b"#" + b"x" * (pad - 1) + b"\n" + block(3, 50) + b" " * 40 + b"\n")



def test_maf_config3_full_size_properties(gpu):
    """stat + call walks over 2 000 000 MAF blocks x 1500 columns (3e9 columns): every block's class counts
    (cigar_cat_ext) and run counts of both walks equal what torch derives from the same rows on the device"""
    import torch
    dev = torch.device("cuda", 0)
    L = 1500
    t, q, n = _synthetic_maf_rows(dev, 2_000_000, L, 11)
    rows = torch.cat([t, q])
    tot = n * L
    cols = torch.full((n,), L, dtype=torch.int64, device=dev)
    t_off = torch.arange(n, device=dev, dtype=torch.int64) * L
    q_off = t_off + tot
    strand = (torch.arange(n, device=dev) % 10 == 0).to(torch.uint8)
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    counts = torch.zeros((n, 11), dtype=torch.int64, device=dev)
    run_cnt = torch.zeros(n, dtype=torch.int64, device=dev)
    gpu.maf_pair_stat(n, rows, t_off, q_off, cols, strand, counts=counts, run_cnt=run_cnt)
    crun = torch.zeros(n, dtype=torch.int64, device=dev)
    gpu.maf_call_runs(n, rows, t_off, q_off, cols, run_cnt=crun)
    torch.cuda.synchronize()
    # independent derivation, 200 000 blocks at a time
    step = 200_000
    for a in range(0, n, step):
        b = min(n, a + step)
        tt, qq = t[a * L:b * L].view(-1, L), q[a * L:b * L].view(-1, L)
        eq, tg, qg = tt == qq, tt == 45, qq == 45
        cls = torch.where(eq, 0, torch.where(tg, 1, torch.where(qg, 2, 3)))           # cigar_cat_ext
        c = counts[a:b]
        neg = strand[a:b].bool()
        assert bool((c[:, 0] == (cls == 0).sum(1)).all()) and bool((c[:, 1] == (cls == 3).sum(1)).all())
        ins_bp, del_bp = (cls == 1).sum(1), (cls == 2).sum(1)
        assert bool((c[:, 3] + c[:, 7] == ins_bp).all()) and bool((c[:, 5] + c[:, 9] == del_bp).all())
        assert bool((c[~neg][:, 6:10] == 0).all()) and bool((c[neg][:, 2:6] == 0).all())
        start = torch.ones_like(cls, dtype=torch.bool)
        start[:, 1:] = cls[:, 1:] != cls[:, :-1]
        assert bool((run_cnt[a:b] == start.sum(1)).all())
        assert bool((c[:, 2] + c[:, 6] == (start & (cls == 1)).sum(1)).all())      # insertion events
        assert bool((c[:, 4] + c[:, 8] == (start & (cls == 2)).sum(1)).all())
        ccls = torch.where(tg & qg, 4, torch.where(tg, 1, torch.where(qg, 2, torch.where(eq, 0, 3))))  # caller classes
        cstart = torch.ones_like(ccls, dtype=torch.bool)
        cstart[:, 1:] = ccls[:, 1:] != ccls[:, :-1]
        assert bool((crun[a:b] == cstart.sum(1)).all())
    # ... and against the ORACLE itself on a sample of the 2 000 000 blocks (orc_parse_maf_seq_to_cigar, cigar.rs:344-432: the
    # eleven counters and the run count of the cg:Z: text), so that the at-size check does not rest on torch expectations alone
    ch, rh = counts.cpu().numpy(), run_cnt.cpu().numpy()
    for i in list(range(0, n, n // 120)) + [n - 1]:
        tb_, qb_ = t[(i % 200_000) * L:(i % 200_000 + 1) * L].cpu().numpy().tobytes(), q[(i % 200_000) * L:(i % 200_000 + 1) * L].cpu().numpy().tobytes()
        exp_counts, exp_txt = orc.parse_maf_seq_to_cigar(tb_, qb_, int(strand[i]))
        assert tuple(int(x) for x in ch[i]) == exp_counts, i
        assert int(rh[i]) == sum(1 for c in exp_txt if not c.isdigit()), i
    # EXACT on a 200 000-block slice: the whole run list of the caller walk — every run's start column, class and the non-gap
    # target / query characters in front of it (what the event rules of caller.rs:444-608 read) — against torch
    n2 = 200_000
    off = torch.zeros(n2 + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(crun[:n2], 0)
    nrun = int(off[-1])
    runs = torch.zeros(3 * nrun + 3, dtype=torch.int64, device=dev)
    gpu.maf_call_runs(n2, rows, t_off[:n2].contiguous(), q_off[:n2].contiguous(), cols[:n2].contiguous(), runs=runs, run_off=off)
    torch.cuda.synchronize()
    exp = torch.empty(3 * nrun, dtype=torch.int64, device=dev)
    done = 0
    for a in range(0, n2, 50_000):
        b = min(n2, a + 50_000)
        tt, qq = t[a * L:b * L].view(-1, L), q[a * L:b * L].view(-1, L)
        eq, tg, qg = tt == qq, tt == 45, qq == 45
        ccls = torch.where(tg & qg, 4, torch.where(tg, 1, torch.where(qg, 2, torch.where(eq, 0, 3))))
        cstart = torch.ones_like(ccls, dtype=torch.bool)
        cstart[:, 1:] = ccls[:, 1:] != ccls[:, :-1]
        tb = torch.cumsum((~tg).to(torch.int64), 1) - (~tg).to(torch.int64)
        qb = torch.cumsum((~qg).to(torch.int64), 1) - (~qg).to(torch.int64)
        idx = cstart.flatten().nonzero().flatten()
        k = idx.numel()
        col = idx % L
        e = exp[3 * done:3 * (done + k)].view(-1, 3)
        e[:, 0] = (col << 3) | ccls.flatten()[idx]
        e[:, 1] = tb.flatten()[idx]
        e[:, 2] = qb.flatten()[idx]
        done += k
        del tt, qq, eq, tg, qg, ccls, cstart, tb, qb, idx
    assert done == nrun and bool(torch.equal(exp, runs[:3 * nrun])), "run list differs from the torch expectation"


def test_maf_call_vcf_at_size(gpu):
    """`call -s -i -l 2` on 200 000 MAF blocks x 1 500 columns through K4 + K19 (wga_maf_call_vcf: chunk cuts, after_m rules,
    rows): the number of rows of every kind against what torch derives from the same rows on the device (a SNP row per X
    column; an INS / DEL row per target- / query-gap run longer than the cutoff whose nearest earlier run that is not
    both-gap is '=' or X; an <INV> row per '-' block), every row's 8 tab-separated columns in place, and a sample of blocks
    byte for byte against the oracle (orc_call_var_maf_record: caller.rs:115-265,388-608)"""
    import torch
    dev = torch.device("cuda", 0)
    L, n, svlen = 1500, 200_000, 2
    t, q, n = _synthetic_maf_rows(dev, n, L, 23)
    rows = torch.cat([t, q])
    tot = n * L
    cols = torch.full((n,), L, dtype=torch.int64, device=dev)
    t_off = torch.arange(n, device=dev, dtype=torch.int64) * L
    q_off = t_off + tot
    neg = (torch.arange(n, device=dev) % 10 == 0)
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    crun = torch.zeros(n, dtype=torch.int64, device=dev)
    gpu.maf_call_runs(n, rows, t_off, q_off, cols, run_cnt=crun)
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(crun, 0)
    nrun = int(off[-1])
    runs = torch.zeros(3 * nrun + 3, dtype=torch.int64, device=dev)
    gpu.maf_call_runs(n, rows, t_off, q_off, cols, run_cnt=crun, runs=runs, run_off=off)
    names = b"ref.chr1qry.chr1\0"
    recs = np.zeros(n, dtype=engine.MAF_VCF_REC_DTYPE)
    recs["t_name_off"], recs["t_name_len"], recs["q_name_off"], recs["q_name_len"] = 0, 8, 8, 8
    recs["t_start"] = 1600 * np.arange(n, dtype=np.uint64)
    recs["q_start"] = 1700 * np.arange(n, dtype=np.uint64)
    recs["q_size"] = 4_000_000_000
    recs["q_neg"] = neg.cpu().numpy().astype(np.uint32)
    d_recs = torch.from_numpy(recs.view(np.uint8).reshape(n, -1).copy()).to(dev)
    d_names = torch.tensor(list(names), dtype=torch.uint8, device=dev)
    nb = torch.zeros(n, dtype=torch.int64, device=dev)
    err = torch.zeros((n, 2), dtype=torch.int64, device=dev)
    gpu.maf_call_vcf(n, rows, t_off, q_off, cols, runs, off, d_recs, d_names, True, True, svlen, 1_000_000, nbytes=nb, err=err)
    torch.cuda.synchronize()
    assert bool((err[:, 0] == -1).all())
    toff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    toff[1:] = torch.cumsum(nb, 0)
    n_text = int(toff[-1])
    text = torch.zeros(n_text + 64, dtype=torch.uint8, device=dev)
    gpu.maf_call_vcf(n, rows, t_off, q_off, cols, runs, off, d_recs, d_names, True, True, svlen, 1_000_000, out=text, out_off=toff)
    torch.cuda.synchronize()
    text = text[:n_text]
    # rows of every kind, from the columns themselves
    tt, qq = t.view(n, L), q.view(n, L)
    tg, qg = tt == 45, qq == 45
    cls = torch.where(tg & qg, 4, torch.where(tg, 1, torch.where(qg, 2, torch.where(tt == qq, 0, 3))))
    n_snp = int((cls == 3).sum())
    r3 = runs[:3 * nrun].view(nrun, 3)
    rcls = r3[:, 0] & 7
    rstart = r3[:, 0] >> 3
    blk = torch.repeat_interleave(torch.arange(n, device=dev), crun)
    rend = torch.empty_like(rstart)
    rend[:-1] = rstart[1:]
    last = torch.zeros(nrun, dtype=torch.bool, device=dev)
    last[off[1:] - 1] = True
    rend[last] = L
    rlen = rend - rstart
    # the nearest earlier run that is not both-gap, inside the block: a running maximum over the indices of such runs
    idx = torch.arange(nrun, device=dev)
    nonw = torch.where(rcls != 4, idx, torch.full_like(idx, -1))
    prev = torch.cummax(nonw, 0).values
    prev_excl = torch.empty_like(prev)
    prev_excl[0] = -1
    prev_excl[1:] = prev[:-1]
    ok_prev = (prev_excl >= 0) & (blk[prev_excl.clamp(min=0)] == blk) & ((rcls[prev_excl.clamp(min=0)] == 0) | (rcls[prev_excl.clamp(min=0)] == 3))
    n_sv = int((((rcls == 1) | (rcls == 2)) & (rlen > svlen) & ok_prev).sum())
    n_negb = int((neg & ((~tg).sum(1) > 0)).sum())
    nl = int((text == 10).sum())
    assert int((text == 9).sum()) == 9 * nl, (int((text == 9).sum()), nl, n_text)   # ten columns a row
    host = text.cpu().numpy().tobytes()
    # an <INV> row per chunk of a '-' block that holds a target base: a block is cut behind its last gap segment of `svlen`
    # columns (caller.rs:186-216), so one or two per block — the sample below pins the cuts exactly
    n_inv = host.count(b"SVTYPE=INV")
    assert n_negb <= n_inv <= 2 * n_negb
    assert host.count(b"SVTYPE=INS") + host.count(b"SVTYPE=DEL") == n_sv and n_sv > 1000
    assert nl == n_snp + n_sv + n_inv, (nl, n_snp, n_sv, n_inv)
    th, qh, tof = t.cpu().numpy(), q.cpu().numpy(), toff.cpu().numpy()
    for i in list(range(0, n, n // 150)) + [n - 1]:
        want = orc.call_var_maf_record("ref.chr1", "qry.chr1", th[i * L:(i + 1) * L].tobytes(), qh[i * L:(i + 1) * L].tobytes(),
                                       int(recs["t_start"][i]), int(recs["q_start"][i]), int((qh[i * L:(i + 1) * L] != 45).sum()),
                                       4_000_000_000, bool(recs["q_neg"][i]), True, True, svlen, 1_000_000)
        assert host[int(tof[i]):int(tof[i + 1])].decode() == want, i
    gpu.reset_stream()


def test_maf_walks_ragged_blocks_exact(gpu):
    """150 000 MAF blocks of 1 .. 3 000 columns (2.2e8 columns; neighbours of every length mix: the walks take two short blocks
    as one column stream where that saves a step, single walks otherwise): every counter of the stat walk, and BOTH walks' whole
    run lists — start column, class, and for the caller walk the non-gap target / query characters in front — against torch"""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    n = 150_000
    cols = torch.randint(1, 3001, (n,), device=dev, generator=g, dtype=torch.int64)
    cols[::1000] = 0                                            # empty blocks between the others
    cols[7::5000] = 1024
    cols[9::5000] = 16
    tot = int(cols.sum())
    t, q, _ = _synthetic_maf_rows(dev, 1, tot, 12)
    t[::100003] = 0xAD                                          # not text: the byte tests' exact path in those steps
    rows = torch.cat([t, q])
    t_off = torch.cumsum(cols, 0) - cols
    q_off = t_off + tot
    strand = (torch.arange(n, device=dev) % 7 == 0).to(torch.uint8)
    blk = torch.repeat_interleave(torch.arange(n, device=dev), cols)
    col = torch.arange(tot, device=dev) - t_off[blk]
    eq, tg, qg = t == q, t == 45, q == 45
    first = col == 0
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        # ---- the stat walk: counters, run counts, run list ----
        counts = torch.zeros((n, 11), dtype=torch.int64, device=dev)
        run_cnt = torch.zeros(n, dtype=torch.int64, device=dev)
        gpu.maf_pair_stat(n, rows, t_off, q_off, cols, strand, counts=counts, run_cnt=run_cnt)
        cls = torch.where(eq, 0, torch.where(tg, 1, torch.where(qg, 2, 3)))
        start = first.clone()
        start[1:] |= cls[1:] != cls[:-1]
        per = lambda m: torch.bincount(blk[m], minlength=n)
        neg = strand.bool()
        z = torch.zeros(n, dtype=torch.int64, device=dev)
        want = torch.stack([per(cls == 0), per(cls == 3),
                            torch.where(neg, z, per(start & (cls == 1))), torch.where(neg, z, per(cls == 1)),
                            torch.where(neg, z, per(start & (cls == 2))), torch.where(neg, z, per(cls == 2)),
                            torch.where(neg, per(start & (cls == 1)), z), torch.where(neg, per(cls == 1), z),
                            torch.where(neg, per(start & (cls == 2)), z), torch.where(neg, per(cls == 2), z),
                            neg.to(torch.int64)], 1)
        assert bool(torch.equal(counts, want)), "stat counters differ"
        assert bool(torch.equal(run_cnt, per(start)))
        off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(run_cnt, 0)
        runs = torch.zeros(int(off[-1]) + 1, dtype=torch.int64, device=dev)
        gpu.maf_pair_stat(n, rows, t_off, q_off, cols, strand, counts=counts, run_cnt=run_cnt, runs=runs, run_off=off)
        idx = start.nonzero().flatten()
        assert bool(torch.equal(runs[:-1], (col[idx] << 3) | cls[idx])), "stat run list differs"
        assert bool(torch.equal(counts, want))
        # ---- the caller walk ----
        crun = torch.zeros(n, dtype=torch.int64, device=dev)
        gpu.maf_call_runs(n, rows, t_off, q_off, cols, run_cnt=crun)
        ccls = torch.where(tg & qg, 4, torch.where(tg, 1, torch.where(qg, 2, torch.where(eq, 0, 3))))
        cstart = first.clone()
        cstart[1:] |= ccls[1:] != ccls[:-1]
        assert bool(torch.equal(crun, per(cstart)))
        coff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        coff[1:] = torch.cumsum(crun, 0)
        cruns = torch.zeros(3 * int(coff[-1]) + 3, dtype=torch.int64, device=dev)
        gpu.maf_call_runs(n, rows, t_off, q_off, cols, run_cnt=crun, runs=cruns, run_off=coff)
        tn, qn = (~tg).to(torch.int64), (~qg).to(torch.int64)
        tcs, qcs = torch.cumsum(tn, 0) - tn, torch.cumsum(qn, 0) - qn          # non-gap characters in front, over all blocks ...
        nz = (cols > 0).nonzero().flatten()
        tb0, qb0 = torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
        tb0[nz], qb0[nz] = tcs[t_off[nz]], qcs[t_off[nz]]                        # ... minus what stands in front of the block
        cidx = cstart.nonzero().flatten()
        exp = torch.stack([(col[cidx] << 3) | ccls[cidx], tcs[cidx] - tb0[blk[cidx]], qcs[cidx] - qb0[blk[cidx]]], 1).flatten()
        assert bool(torch.equal(cruns[:exp.numel()], exp)), "caller run list differs"
    finally:
        gpu.reset_stream()


def test_pafcov_config4_scaled_properties(gpu):
    """config 4 shaped coverage (8 targets x 12.5 Mb here, 60 000 records): per target, the summed coverage equals
    the M/= bases K1 counts for its records; coverage is never negative nor above the number of records"""
    import torch
    dev = torch.device("cuda", 0)
    tb = synth.make_paf_batch_torch(23, 60_000, 1300, 12_500_000, dev)
    n = tb["n"]
    nt, tlen = 8, int(tb["t_pool"].numel())
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], n, tb["n_ops"])
    target_id = (torch.arange(n, device=dev) % nt).to(torch.int32)
    cov_len = torch.full((nt,), tlen, dtype=torch.int64, device=dev)
    cov_off = torch.arange(nt, device=dev, dtype=torch.int64) * (tlen + 4)
    total = int(nt * (tlen + 4))
    cov = torch.zeros(total + 8, dtype=torch.int32, device=dev)
    gpu.pafcov_accumulate(batch, target_id, tb["t_src_off"], cov_off, cov_len, cov, total)
    gpu.pafcov_finalize(nt, cov_off, cov_len, cov)
    counts = torch.zeros((n, 11), dtype=torch.int64, device=dev)
    diag = torch.zeros((n, 3), dtype=torch.int64, device=dev)
    gpu.cigar_stat(batch, counts, diag, None)
    torch.cuda.synchronize()
    # M and = bases per record: match count minus nothing (X is "mismatch"); every record lies inside its target
    m_eq = counts[:, 0]
    for k in range(nt):
        c = cov[int(cov_off[k]):int(cov_off[k]) + tlen].long()
        assert int(c.sum()) == int(m_eq[target_id == k].sum()), k
        assert int(c.min()) >= 0 and int(c.max()) <= n
    assert int(cov[total:].abs().sum()) == 0


def test_pafpseudo_config5_long_cigar_cross_check(gpu):
    """>= 200 kop records: the pseudo-MAF row (target coordinates) must equal paf2maf's query row with the
    columns where the target row is gapped removed — two kernels, one answer"""
    import torch
    from wgatools_amd import pipeline
    dev = torch.device("cuda", 0)
    tb = synth.make_paf_batch_torch(31, 24, 250_000, 40_000_000, dev, sigma=0.05)
    n = tb["n"]
    job = pipeline.Paf2MafStatJob(gpu, tb)
    job.bind_stream()
    job.step()
    batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], n, tb["n_ops"])
    seg = tb["mx"] + tb["d"]
    dst_off = torch.zeros(n, dtype=torch.int64, device=dev)
    dst_off[1:] = torch.cumsum(seg, 0)[:-1]
    total = int(seg.sum())
    out = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
    skip = torch.zeros(n, dtype=torch.int64, device=dev)
    gpu.pafpseudo_fill(batch, 1, tb["q_pool"], int(tb["q_pool"].numel()), tb["q_src_off"], tb["q_src_len"], skip, out, dst_off)
    torch.cuda.synchronize()
    for i in range(n):
        L = int(tb["mx"][i] + tb["i"][i] + tb["d"][i])
        to, qo = int(job.t_row_off[i]), int(job.q_row_off[i])
        trow, qrow = job.out[to:to + L], job.out[qo:qo + L]
        want = qrow[trow != 45]
        got = out[int(dst_off[i]):int(dst_off[i]) + int(seg[i])]
        assert want.numel() == got.numel() and bool((want == got).all()), i


def test_pafcov_many_small_targets(gpu):
    pc.check_pafcov_many_small_targets(gpu, nt=400)
    pc.check_pafcov_many_small_targets(gpu, seed=6, nt=3000)


def test_pafcov_one_long_target_both_protocols(gpu):
    """ONE target of 2e9 counters (244 141 windows): in the counting replay every window hangs on the window in front of it — the
    longest chain the look-back can meet, a thousand consecutive windows of it in flight at a time — and the array lies beyond
    2^31 bytes.  2 000 000 records; accumulate + finalize and accumulate_final give the same counters, and those are the
    running sum of torch's +1 / -1 marks"""
    import torch
    dev = torch.device("cuda", 0)
    tlen, n = 2_000_000_000, 2_000_000
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    tb = synth.make_paf_batch_torch(431, n, 1300, tlen, dev)
    batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], n, tb["n_ops"])
    cov_off = torch.tensor([3], dtype=torch.int64, device=dev)          # not aligned to anything
    cov_len = torch.tensor([tlen], dtype=torch.int64, device=dev)
    total = tlen + 3
    tid = torch.zeros(n, dtype=torch.int32, device=dev)
    cov = torch.zeros(total + 8, dtype=torch.int32, device=dev)
    gpu.pafcov_accumulate(batch, tid, tb["t_src_off"], cov_off, cov_len, cov, total)
    gpu.pafcov_finalize(1, cov_off, cov_len, cov)
    one = torch.zeros(total + 8, dtype=torch.int32, device=dev)
    gpu.pafcov_accumulate_final(batch, tid, tb["t_src_off"], cov_off, cov_len, 1, one, total)
    torch.cuda.synchronize()
    assert bool(torch.equal(cov, one))
    del one
    exp = torch.zeros(total + 8, dtype=torch.int32, device=dev)
    op_off, sub = tb["op_off"], 200_000
    for r0 in range(0, n, sub):                     # 200 000 records at a time (the whole op stream as int64 would not fit)
        r1 = min(n, r0 + sub)
        a, b = int(op_off[r0]), int(op_off[r1])
        o = tb["ops"][a:b].to(torch.int64) & 0xFFFFFFFF
        code, ln = o & 15, o >> 4
        nper = op_off[r0 + 1:r1 + 1] - op_off[r0:r1]
        rec = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), nper)
        adv = torch.where((code == 1) | (code == 4) | (code == 9), torch.zeros_like(ln), ln)
        cs = torch.cumsum(adv, 0)
        start_cs = (cs - adv)[op_off[r0:r1] - a]
        pos = tb["t_src_off"][r0:r1][rec] + (cs - adv) - start_cs[rec]
        cover = (code == 0) | (code == 7)
        p0, p1 = pos[cover], (pos + ln)[cover]
        m0, m1 = p0 < tlen, p1 < tlen
        exp.index_add_(0, 3 + p0[m0], torch.ones(int(m0.sum()), dtype=torch.int32, device=dev))
        exp.index_add_(0, 3 + p1[m1], torch.full((int(m1.sum()),), -1, dtype=torch.int32, device=dev))
        del o, code, ln, rec, adv, cs, pos, cover, p0, p1, m0, m1
    e = torch.cumsum(exp[3:3 + tlen], 0, dtype=torch.int32)
    assert bool(torch.equal(e, cov[3:3 + tlen])) and int(cov[:3].abs().sum()) == 0 and int(cov[3 + tlen:].abs().sum()) == 0
    gpu.reset_stream()


def test_pafcov_config4_at_stated_size(gpu):
    """BASELINE configs[3] at its stated size: 64 targets x 100 Mb = 6.4e9 int32 counters (25.6 GB), 20 M records of ~1300
    ops generated on the device in ten chunks and gathered into ONE resident batch (2.6e10 ops, 104 GB), accumulated into
    the resident coverage array with ONE wga_pafcov_accumulate call — the array is read and written once, not once per
    chunk (288 GB of HBM hold all of it).  Per target the summed coverage equals the M / = bases K1 counts for the
    records that hit it, coverage is never negative."""
    import torch
    dev = torch.device("cuda", 0)
    nt, tlen, chunks, per = 64, 100_000_000, 10, 2_000_000
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    cov_len = torch.full((nt,), tlen, dtype=torch.int64, device=dev)
    cov_off = torch.arange(nt, device=dev, dtype=torch.int64) * (tlen + 4)
    total = int(nt * (tlen + 4))
    cov = torch.zeros(total + 8, dtype=torch.int32, device=dev)
    want = torch.zeros(nt, dtype=torch.int64, device=dev)
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
        n = tb["n"]
        assert n == per and n_ops + tb["n_ops"] <= cap
        batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], n, tb["n_ops"])
        g = torch.Generator(device=dev)
        g.manual_seed(900 + k)
        tid = torch.randint(0, nt, (n,), device=dev, generator=g, dtype=torch.int32)
        counts = torch.zeros((n, 11), dtype=torch.int64, device=dev)
        diag = torch.zeros((n, 3), dtype=torch.int64, device=dev)
        gpu.cigar_stat(batch, counts, diag, None)
        torch.cuda.synchronize()
        want.index_add_(0, tid.long(), counts[:, 0])
        ops[n_ops:n_ops + tb["n_ops"]] = tb["ops"]
        op_off[k * per + 1:(k + 1) * per + 1] = tb["op_off"][1:] + n_ops
        strand[k * per:(k + 1) * per] = tb["strand_neg"]
        t_start[k * per:(k + 1) * per] = tb["t_src_off"]
        target_id[k * per:(k + 1) * per] = tid
        n_ops += tb["n_ops"]
        del tb, batch, counts, diag, tid
        torch.cuda.empty_cache()
    batch = engine.Batch(ops, op_off, strand, n_all, n_ops)
    # a first, untimed call sizes the context's work lists (gigabytes of device allocations, seconds when HBM is this full);
    # a long-lived caller pays them once
    gpu.pafcov_accumulate(batch, target_id, t_start, cov_off, cov_len, cov, total)
    cov.zero_()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    gpu.pafcov_accumulate(batch, target_id, t_start, cov_off, cov_len, cov, total)
    e1.record()
    gpu.pafcov_finalize(nt, cov_off, cov_len, cov)
    e2.record()
    torch.cuda.synchronize()
    ms_acc, ms_fin = e0.elapsed_time(e1), e1.elapsed_time(e2)
    print("\nconfig 4 at size: %d records, %.3g ops, %d x %d counters: ONE accumulate call %.1f ms (%.0f GB/s of op stream), "
          "finalize %.1f ms (%.0f GB/s over 8 B per counter)" % (n_all, n_ops, nt, tlen, ms_acc, 4 * n_ops / ms_acc / 1e6,
                                                                ms_fin, 8.0 * nt * tlen / ms_fin / 1e6))
    # the same job as ONE call (marks and the marks -> counts scan in one pass over the array): every counter as above
    cov_f = torch.zeros(total + 8, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    e0.record()
    gpu.pafcov_accumulate_final(batch, target_id, t_start, cov_off, cov_len, nt, cov_f, total)
    e1.record()
    torch.cuda.synchronize()
    ms_one = e0.elapsed_time(e1)
    print("config 4 at size: accumulate_final %.1f ms = %.0f GB/s over 4 B per op + 8 B per counter" % (
        ms_one, (4.0 * n_ops + 8.0 * nt * tlen) / ms_one / 1e6))
    assert bool(torch.equal(cov, cov_f)), "accumulate_final differs from accumulate + finalize"
    del cov_f
    torch.cuda.empty_cache()
    for t in range(nt):
        c = cov[int(cov_off[t]):int(cov_off[t]) + tlen]
        assert int(c.sum(dtype=torch.int64)) == int(want[t]), t
        assert int(c.min()) >= 0
    # EXACT: every one of the 6.4e9 counters against an expectation computed with torch alone (update_cov_vec,
    # cigar.rs:720-733: M / = ops cover [pos, pos + len) below the target's length; I / S do not move, everything else moves):
    # +1 / -1 marks of every M / = op by index_add_, 200 000 records at a time, then a running sum per target
    exp = torch.zeros(total + 8, dtype=torch.int32, device=dev)
    sub = 200_000
    for r0 in range(0, n_all, sub):
        r1 = min(n_all, r0 + sub)
        a, b = int(op_off[r0]), int(op_off[r1])
        o = ops[a:b].to(torch.int64) & 0xFFFFFFFF
        code, ln = o & 15, o >> 4
        nper = op_off[r0 + 1:r1 + 1] - op_off[r0:r1]
        rec = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), nper)
        stay = (code == 1) | (code == 4) | (code == 9)
        adv = torch.where(stay, torch.zeros_like(ln), ln)
        cs = torch.cumsum(adv, 0)
        start_cs = torch.zeros(r1 - r0, dtype=torch.int64, device=dev)
        first = (op_off[r0:r1] - a)
        has = nper > 0
        start_cs[has] = (cs - adv)[first[has]]
        pos = t_start[r0:r1][rec] + (cs - adv) - start_cs[rec]
        cover = (code == 0) | (code == 7)
        base = cov_off[target_id[r0:r1].long()][rec]
        p0, p1 = pos[cover], (pos + ln)[cover]
        bc = base[cover]
        m0, m1 = p0 < tlen, p1 < tlen
        exp.index_add_(0, (bc + p0)[m0], torch.ones(int(m0.sum()), dtype=torch.int32, device=dev))
        exp.index_add_(0, (bc + p1)[m1], torch.full((int(m1.sum()),), -1, dtype=torch.int32, device=dev))
        del o, code, ln, rec, stay, adv, cs, pos, cover, base, p0, p1, bc, m0, m1
    torch.cuda.empty_cache()
    for t in range(nt):
        lo = int(cov_off[t])
        e = torch.cumsum(exp[lo:lo + tlen], 0, dtype=torch.int32)
        assert bool(torch.equal(e, cov[lo:lo + tlen])), "target %d: coverage differs from the torch expectation" % t
        del e
    gpu.reset_stream()


def test_pafpseudo_stream_equals_block_kernel_at_scale(gpu):
    """30 000 records x mean 5 kop with trimmed heads of every size (none, a few columns, thousands, the whole row): the rows of
    the streaming kernel ("pseudo_variant" 3) and of the block kernel (0) are the same bytes in both modes, nothing is written
    outside the segments, and the stream leaves only the records at the pool's edges to the block kernel"""
    import torch
    dev = torch.device("cuda", 0)
    tb = synth.make_paf_batch_torch(77, 30_000, 5000, 50_000_000, dev)
    n = tb["n"]
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], n, tb["n_ops"])
        seg = tb["mx"] + tb["d"]
        g = torch.Generator(device=dev)
        g.manual_seed(3)
        kind = torch.randint(0, 4, (n,), device=dev, generator=g)
        skip = torch.where(kind == 0, torch.zeros_like(seg),
                           torch.where(kind == 1, torch.randint(1, 40, (n,), device=dev, generator=g),
                                       torch.where(kind == 2, torch.randint(1000, 9000, (n,), device=dev, generator=g), seg)))
        skip = torch.minimum(skip, seg)
        left = seg - skip
        dst_off = torch.zeros(n, dtype=torch.int64, device=dev)
        dst_off[1:] = torch.cumsum(left + 3, 0)[:-1]                      # three bytes between the segments
        total = int((left + 3).sum())
        outs = {}
        for mode in (1, 0):
            for variant in (3, 0):
                gpu.set_param("pseudo_variant", variant)
                out = torch.full((total + 64,), 0x23, dtype=torch.uint8, device=dev)
                diag = gpu.pafpseudo_fill(batch, mode, tb["q_pool"], int(tb["q_pool"].numel()), tb["q_src_off"], tb["q_src_len"], skip, out, dst_off)
                torch.cuda.synchronize()
                outs[(mode, variant)] = out
                if variant == 3:
                    leftover = gpu.get_param("pseudo_stream_left_to_blocks")
                    assert leftover < 50 if mode else leftover == 0, (mode, leftover)
                del diag
            assert bool(torch.equal(outs[(mode, 3)], outs[(mode, 0)])), mode
            covered = torch.zeros(total + 64, dtype=torch.bool, device=dev)
            idx = torch.repeat_interleave(dst_off, left) + (torch.arange(int(left.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(left, 0) - left, left))
            covered[idx] = True
            assert bool((outs[(mode, 3)][~covered] == 0x23).all()) and bool((outs[(mode, 3)][covered] != 0x23).all()), mode
            del covered, idx
    finally:
        gpu.set_param("pseudo_variant", 3)
        gpu.reset_stream()


def _pafpseudo_config5_at_stated_size(gpu):
    """10 000 records of >= 200 kop (2.5e9 ops, 3.7e10 columns) in chunks of 400: the base-mode pseudo-MAF row equals
    paf2maf's query row minus the columns where its target row is gapped (K6 against K2 — the rows of v1, the block kernel:
    K6's rows come from the streaming kernel, so the two share no row code), and equals the block kernel's K6 rows
    ("pseudo_variant" 0); the symbol-mode row equals the op symbols expanded on the device with torch"""
    import torch
    from wgatools_amd import pipeline
    dev = torch.device("cuda", 0)
    per, chunks = 400, 25
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    gpu.set_param("expand_variant", 0)
    sym = torch.zeros(16, dtype=torch.uint8, device=dev)
    sym[0] = sym[7] = ord("1")
    sym[8] = ord("0")
    sym[2] = ord("-")
    ms = {0: 0.0, 1: 0.0}
    out_bytes, n_ops = 0, 0
    for k in range(chunks):
        tb = synth.make_paf_batch_torch(700 + k, per, 250_000, 60_000_000, dev, sigma=0.05)
        n = tb["n"]
        assert int(torch.diff(tb["op_off"]).min()) >= 200_000
        job = pipeline.Paf2MafStatJob(gpu, tb)
        job.bind_stream()
        job.step()
        batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], n, tb["n_ops"])
        seg = tb["mx"] + tb["d"]
        dst_off = torch.zeros(n, dtype=torch.int64, device=dev)
        dst_off[1:] = torch.cumsum(seg, 0)[:-1]
        total = int(seg.sum())
        skip = torch.zeros(n, dtype=torch.int64, device=dev)
        outs = {}
        for mode in (1, 0):
            out = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
            gpu.pafpseudo_fill(batch, mode, tb["q_pool"], int(tb["q_pool"].numel()), tb["q_src_off"], tb["q_src_len"], skip, out, dst_off)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gpu.pafpseudo_fill(batch, mode, tb["q_pool"], int(tb["q_pool"].numel()), tb["q_src_off"], tb["q_src_len"], skip, out, dst_off)
            e1.record()
            torch.cuda.synchronize()
            ms[mode] += e0.elapsed_time(e1)
            outs[mode] = out
        out_bytes += total
        n_ops += tb["n_ops"]
        # base mode against K2: rows are laid out t row, q row per record; concatenated over the chunk
        L = tb["mx"] + tb["i"] + tb["d"]
        rec = torch.repeat_interleave(torch.arange(n, device=dev), L)
        col = torch.arange(int(L.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(L, 0) - L, L)
        trow = job.out[job.t_row_off[rec] + col]
        qrow = job.out[job.q_row_off[rec] + col]
        want = qrow[trow != 45]
        assert want.numel() == total and bool((want == outs[1][:total]).all()), k
        del rec, col, trow, qrow, want
        assert gpu.get_param("pseudo_variant") == 3
        if k % 5 == 0:   # ... and against the block kernel's rows of the same call
            gpu.set_param("pseudo_variant", 0)
            blk = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
            gpu.pafpseudo_fill(batch, 1, tb["q_pool"], int(tb["q_pool"].numel()), tb["q_src_off"], tb["q_src_len"], skip, blk, dst_off)
            gpu.set_param("pseudo_variant", 3)
            assert bool(torch.equal(blk, outs[1])), k
            del blk
        # symbol mode against the ops expanded with torch
        code = (tb["ops"] & 15).long()
        ln = (tb["ops"] >> 4).long()
        keep = (code == 0) | (code == 7) | (code == 8) | (code == 2)
        want_sym = torch.repeat_interleave(sym[code[keep]], ln[keep])
        assert want_sym.numel() == total and bool((want_sym == outs[0][:total]).all()), k
        del tb, job, batch, outs, code, ln, keep, want_sym
        torch.cuda.empty_cache()
    print("\nconfig 5 stress at size: %d records, %.3g ops, %.3g row bytes per mode: base mode %.1f ms in all (%.0f GB/s of "
          "4n + 2 x row bytes), symbol mode %.1f ms (%.0f GB/s of 4n + row bytes)" % (
              per * chunks, n_ops, out_bytes, ms[1], (4 * n_ops + 2 * out_bytes) / ms[1] / 1e6, ms[0],
              (4 * n_ops + out_bytes) / ms[0] / 1e6))


def test_pafpseudo_config5_at_stated_size(gpu):
    try:
        _pafpseudo_config5_at_stated_size(gpu)
    finally:   # the engine is the session's
        gpu.set_param("expand_variant", -1)
        gpu.set_param("pseudo_variant", 3)
        gpu.reset_stream()


def test_fasta_pool(gpu):
    """device-built sequence pools == the host faidx reader's, on multi-contig, ragged-line, CRLF and odd inputs"""
    for t in pc.FASTA_CASES:
        pc.check_fasta_pool(gpu, t)
    rng = np.random.default_rng(3)
    for k in range(8):
        pc.check_fasta_pool(gpu, pc.random_fasta(rng, int(rng.integers(1, 40)), 200_000, crlf=bool(k & 1)))
    pc.check_fasta_pool(gpu, pc.random_fasta(rng, 5, 20_000_000, width=60))


def test_bgzf_inflate(gpu):
    pc.check_bgzf_inflate(gpu)


def test_bgzf_deflate(gpu):
    pc.check_bgzf_deflate(gpu)


def test_bgzf_deflate_at_size_round_trip(gpu):
    """K18 on 2 GB of alignment-row-like text made in HBM (61 035 full members and a ragged one, unaligned input and output):
    every member is taken back by the device inflate (K17) and the bytes are the input's; zlib inflates a sample of members
    and checks their CRC-32 and ISIZE; the stream is about a third of the input."""
    import struct
    import time
    import zlib
    import torch
    dev = torch.device("cuda", 0)
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 2_000_000_123
    g = torch.Generator(device=dev)
    g.manual_seed(18)
    alphabet = torch.tensor(list(b"ACGTACGTACGTACGTACGTACGT----acgtN\n"), dtype=torch.uint8, device=dev)
    text = torch.empty(n + 5 + 16, dtype=torch.uint8, device=dev)
    for a in range(0, n, 1 << 28):
        m = min(1 << 28, n - a)
        text[5 + a:5 + a + m] = alphabet[torch.randint(0, len(alphabet), (m,), device=dev, generator=g)]
    cap = int(gpu.lib.wga_bgzf_bound(n))
    out = torch.full((cap + 64,), 0x23, dtype=torch.uint8, device=dev)
    _, used = gpu.bgzf_compress(text, n, out=engine.DeviceArray(gpu, out.data_ptr(), (cap + 64,), np.uint8, owner=False), eof_marker=True,
                                in_offset=5, out_offset=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, used2 = gpu.bgzf_compress(text, n, out=engine.DeviceArray(gpu, out.data_ptr(), (cap + 64,), np.uint8, owner=False), eof_marker=True,
                                 in_offset=5, out_offset=3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert used2 == used and 0.25 * n < used < 0.40 * n
    assert bool((out[:3] == 0x23).all()) and bool((out[3 + used:] == 0x23).all())
    img = out[3:3 + used].cpu().numpy()
    n_members = (n + 32767) // 32768
    rows = np.zeros(n_members, dtype=[("in_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4"), ("out_off", "<u8")])
    p = 0
    for k in range(n_members):
        assert img[p] == 0x1f and img[p + 1] == 0x8b and img[p + 12] == 0x42 and img[p + 13] == 0x43
        bsize = int(img[p + 16]) + 256 * int(img[p + 17]) + 1
        isize = struct.unpack_from("<I", img, p + bsize - 4)[0]
        assert isize == (32768 if k + 1 < n_members else n - 32768 * (n_members - 1))
        rows[k] = (p + 18, bsize - 26, isize, 32768 * k)
        p += bsize
    assert p + 28 == used and img[p:].tobytes() == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    for k in list(range(0, n_members, 4099)) + [n_members - 1]:             # zlib on a sample of members
        a, ln = int(rows["in_off"][k]), int(rows["in_len"][k])
        data = zlib.decompress(img[a:a + ln].tobytes(), -15)
        crc, isize = struct.unpack_from("<II", img, a + ln)
        want = text[5 + 32768 * k:5 + 32768 * k + isize].cpu().numpy().tobytes()
        assert data == want and zlib.crc32(data) & 0xFFFFFFFF == crc
    back = torch.full((n + 64,), 0x23, dtype=torch.uint8, device=dev)
    status = torch.full((n_members,), 0xFF, dtype=torch.int32, device=dev)
    d_rows = torch.from_numpy(rows.view(np.uint8).copy()).to(dev)
    gpu.bgzf_inflate(out[3:], used, n_members, d_rows, back, status)
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0
    assert torch.equal(back[:n], text[5:5 + n]) and bool((back[n:] == 0x23).all())
    print("\nK18 at size: %.2e bytes -> %.2e (%.3f) in %.1f ms = %.0f GB/s of input (plan + scan + emit + the size's trip to the host)"
          % (n, used, used / n, dt * 1e3, n / dt / 1e9))
    gpu.reset_stream()


@pytest.mark.gpu
def test_reduce_scatter_i32_on_hardware():
    """wga_reduce_scatter_i32 (pafcov --spread: the coverage merge of pafcov.rs:29-53 across devices) executed by the GPU: three
    contexts on the one device of this box ("reduce_same_device_ok" lifts the one-context-per-device rule for this test), the
    peers' slices read in place by one kernel per context and, with "reduce_staged", pulled into scratch over two streams
    first; nothing waits on the host inside the call (events order the three streams).  Slice g of buffer g = the sum over
    the contexts, the rest untouched"""
    import ctypes as C
    import torch  # noqa: F401
    from wgatools_amd import build, _lib
    lib = _lib.load(build.HIP_LIB)
    engs = [engine.Engine(0, lib) for _ in range(3)]
    engs[0].set_param("reduce_same_device_ok", 1)
    rng = np.random.default_rng(4)
    try:
        for staged in (0, 1):
            for e in engs:
                e.set_param("reduce_staged", staged)
            for count in (1, 7, 70001, 25_000_000):
                host = [rng.integers(-1000, 1000, count, dtype=np.int32) for _ in range(3)]
                bufs = [e.upload(h) for e, h in zip(engs, host)]
                cx = (C.c_void_p * 3)(*[e.ctx for e in engs])
                bp = (C.c_void_p * 3)(*[b.ptr for b in bufs])
                assert lib.wga_reduce_scatter_i32(cx, 3, bp, count) == 0
                total = host[0].astype(np.int64) + host[1] + host[2]
                for g in range(3):
                    lo, hi = count * g // 3, count * (g + 1) // 3
                    got = bufs[g].numpy()[:count]          # a download on context g's stream: behind the call's events
                    assert (got[lo:hi] == total[lo:hi]).all(), (staged, count, g)
                    mask = np.ones(count, dtype=bool)
                    mask[lo:hi] = False
                    assert (got[mask] == host[g][mask]).all(), (staged, count, g)
                for b in bufs:
                    b.free()
        dup = (C.c_void_p * 2)(engs[0].ctx, engs[0].ctx)
        b2 = (C.c_void_p * 2)(None, None)
        assert lib.wga_reduce_scatter_i32(dup, 2, b2, 0) == -1
    finally:
        for e in engs:
            e.close()


def test_reduce_scatter_i32_across_devices():
    """the same call over the REAL devices of the box (no "reduce_same_device_ok"): one context per device, up to eight — the
    peers' slices cross xGMI, read in place by one kernel per device, or pulled over N - 1 streams with "reduce_staged".  Skips
    on a one-GPU box (the driver's pool has those); on a multi-GPU lease it is the first evidence of bytes over the links, and
    it prints the rate of the 400 MB case"""
    import ctypes as C
    import time
    import torch  # noqa: F401
    from wgatools_amd import build, _lib
    lib = _lib.load(build.HIP_LIB)
    have = int(lib.wga_device_count())
    if have < 2:
        pytest.skip("one device on this box: the cross-device reduce needs two")
    ng = min(have, 8)
    engs = [engine.Engine(g, lib) for g in range(ng)]
    rng = np.random.default_rng(14)
    try:
        for staged in (0, 1):
            for e in engs:
                e.set_param("reduce_staged", staged)
            for count in (1, 70001, 100_000_000):
                host = [rng.integers(-1000, 1000, count, dtype=np.int32) for _ in range(ng)]
                bufs = [e.upload(h) for e, h in zip(engs, host)]
                cx = (C.c_void_p * ng)(*[e.ctx for e in engs])
                bp = (C.c_void_p * ng)(*[b.ptr for b in bufs])
                for e in engs:
                    e.sync()
                t0 = time.perf_counter()
                assert lib.wga_reduce_scatter_i32(cx, ng, bp, count) == 0
                for e in engs:
                    e.sync()
                dt = time.perf_counter() - t0
                total = np.sum(np.stack(host).astype(np.int64), axis=0)
                for g in range(ng):
                    lo, hi = count * g // ng, count * (g + 1) // ng
                    got = bufs[g].numpy()[:count]
                    assert (got[lo:hi] == total[lo:hi]).all(), (staged, count, g)
                    mask = np.ones(count, dtype=bool)
                    mask[lo:hi] = False
                    assert (got[mask] == host[g][mask]).all(), (staged, count, g)
                if count == 100_000_000:
                    moved = 4.0 * count * (ng - 1) / ng        # bytes every device reads from its peers
                    print("\nreduce_scatter_i32 over %d devices (%s): %.2f ms, %.1f GB/s into each device" % (
                        ng, "staged" if staged else "peer reads", dt * 1e3, moved / dt / 1e9))
                for b in bufs:
                    b.free()
    finally:
        for e in engs:
            e.close()


def test_paf2maf_stream_kernel_pools_beyond_4_gb(gpu):
    """sequence pools of 4.6 GB: the streaming kernel addresses a record segment's source through a buffer whose base lies a
    little in front of what the job can reach (32-bit offsets) — slices beyond 2^32 must read the same bytes as v1's 64-bit
    pointers, on both strands; a few records against the oracle"""
    import torch
    from wgatools_amd import pipeline
    dev = torch.device("cuda", 0)
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    tb = synth.make_paf_batch_torch(321, 3000, 3000, 4_600_000_000, dev)
    assert int(tb["t_src_off"].max()) > (1 << 32) and int(tb["q_src_off"].max()) > (1 << 32)
    outs = []
    for v in (0, 3):
        gpu.set_param("expand_variant", v)
        job = pipeline.Paf2MafStatJob(gpu, tb, with_text=True)
        job.out.fill_(0x23)
        job.bind_stream()
        job.step()
        torch.cuda.synchronize()
        assert bool((job.diag == -1).all()) and gpu.get_param("expand_variant_used") == v
        outs.append(job.out)
        if v == 3:
            far = torch.nonzero((tb["q_src_off"] > (1 << 32)) & (tb["t_src_off"] > (1 << 32))).flatten()[:4].tolist()
            for i in far:
                et, eq = pc.oracle_rows(synth.torch_batch_record_to_numpy(tb, i), 0)
                gt, gq = job.record_rows(i)
                assert gt == et and gq == eq, i
        del job
    assert bool(torch.equal(outs[0], outs[1]))
    gpu.set_param("expand_variant", pc.DEFAULT_EXPAND_VARIANT)
    gpu.reset_stream()
