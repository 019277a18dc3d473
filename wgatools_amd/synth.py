"""Synthetic PAF workloads of BASELINE.json's shapes, generated with numpy.

`make_paf_batch` draws the config-2 mixture: records of ~lognormal(mean_ops, sigma 0.5) ops that
alternate a match run (`=` or `M`, len ~ Geom(mean 24)) with an edit (X len 1: 60 %, I: 20 %,
D: 20 %; indel len ~ Geom(mean 3) with a 1 % heavy tail U[50, 2000]); strand 50/50; target and
query slices placed uniformly in two sequence pools of ACGT (+0.1 % N, 5 % lower-case runs).
"""
import numpy as np

OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)
OP_CHARS = "MIDNSHP=X"


def make_pool(rng, nbytes, n_frac=0.001, lower_frac=0.05):
    """uniform ACGT with a sprinkle of N and lower-case runs (exercises case + revcomp)"""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    pool = acgt[rng.integers(0, 4, size=nbytes)]
    if n_frac > 0 and nbytes:
        k = int(nbytes * n_frac)
        pool[rng.integers(0, nbytes, size=k)] = ord("N")
    if lower_frac > 0 and nbytes:
        run = 200
        k = max(1, int(nbytes * lower_frac / run))
        starts = rng.integers(0, max(1, nbytes - run), size=k)
        for s in starts[: 20000]:
            pool[s:s + run] |= 0x20
    return pool


def make_ops(rng, n_rec, mean_ops, sigma=0.5, use_m=False, min_ops=1, max_ops=None):
    """-> ops (u32 packed), op_off (u64, n+1), per-class sums per record"""
    mu = np.log(mean_ops) - 0.5 * sigma * sigma
    n_ops = np.maximum(min_ops, rng.lognormal(mu, sigma, size=n_rec).astype(np.int64))
    if max_ops is not None:
        n_ops = np.minimum(n_ops, max_ops)
    op_off = np.zeros(n_rec + 1, dtype=np.uint64)
    np.cumsum(n_ops, out=op_off[1:])
    total = int(op_off[-1])
    idx_in_rec = np.arange(total, dtype=np.int64) - np.repeat(op_off[:-1].astype(np.int64), n_ops)
    is_match = (idx_in_rec & 1) == 0
    code = np.empty(total, dtype=np.uint32)
    length = np.empty(total, dtype=np.uint32)
    nm = int(is_match.sum())
    code[is_match] = OP_M if use_m else OP_EQ
    length[is_match] = rng.geometric(1.0 / 24.0, size=nm)
    ne = total - nm
    u = rng.random(ne)
    ecode = np.where(u < 0.6, OP_X, np.where(u < 0.8, OP_I, OP_D)).astype(np.uint32)
    elen = rng.geometric(1.0 / 3.0, size=ne).astype(np.uint32)
    heavy = rng.random(ne) < 0.01
    elen[heavy] = rng.integers(50, 2001, size=int(heavy.sum()))
    elen[ecode == OP_X] = 1
    if use_m:
        ecode[ecode == OP_X] = OP_M
    code[~is_match] = ecode
    length[~is_match] = elen
    ops = (length << np.uint32(4)) | code
    return ops.astype(np.uint32), op_off, code, length


def class_sums(code, length, op_off):
    """per-record sums of M+=+X, I, D lengths (numpy reduceat; empty records -> 0)"""
    n = len(op_off) - 1
    out = {}
    starts = op_off[:-1].astype(np.int64)
    for name, mask in (("mx", (code == OP_M) | (code == OP_EQ) | (code == OP_X)),
                       ("i", code == OP_I), ("d", code == OP_D)):
        v = np.where(mask, length, 0).astype(np.uint64)
        c = np.zeros(len(v) + 1, dtype=np.uint64)
        np.cumsum(v, out=c[1:])
        out[name] = c[op_off[1:].astype(np.int64)] - c[starts]
    assert len(out["mx"]) == n
    return out


def make_paf_batch(seed, n_rec, mean_ops, pool_bytes, use_m=False, sigma=0.5, max_ops=None,
                   neg_frac=0.5):
    """A full paf2maf problem: packed CIGARs + slices into two pools (all numpy, host side)."""
    rng = np.random.default_rng(seed)
    ops, op_off, code, length = make_ops(rng, n_rec, mean_ops, sigma, use_m, max_ops=max_ops)
    cs = class_sums(code, length, op_off)
    t_len = cs["mx"] + cs["d"]
    q_len = cs["mx"] + cs["i"]
    need = int(max(t_len.max(initial=0), q_len.max(initial=0))) + 64
    pool_bytes = max(int(pool_bytes), need)
    t_pool = make_pool(rng, pool_bytes)
    q_pool = make_pool(rng, pool_bytes)
    t_off = (rng.random(n_rec) * (pool_bytes - t_len.astype(np.float64))).astype(np.uint64)
    q_off = (rng.random(n_rec) * (pool_bytes - q_len.astype(np.float64))).astype(np.uint64)
    strand = (rng.random(n_rec) < neg_frac).astype(np.uint8)
    return dict(ops=ops, op_off=op_off, strand_neg=strand, t_pool=t_pool, q_pool=q_pool,
                t_src_off=t_off, t_src_len=t_len.astype(np.uint64), q_src_off=q_off,
                q_src_len=q_len.astype(np.uint64), code=code, length=length)


def cigar_text(ops_slice):
    """packed ops of one record -> CIGAR text (no tag)"""
    out = []
    for w in ops_slice.tolist():
        c = w & 15
        ch = OP_CHARS[c] if c < 9 else ("I" if c == 9 else "D" if c == 10 else "B")
        out.append("%d%s" % (w >> 4, ch))
    return "".join(out)


# ------------------------------------------------------------------------------------------------
# device-side generator (torch is plumbing: it only fills HBM with synthetic input)
# ------------------------------------------------------------------------------------------------
def record_lengths(seed, n_rec, mean_ops, sigma=0.5):
    """ops per record of the synthetic mixture (lognormal, at least 1): drawn apart so that a global batch can be cut
    into shards whose records keep their lengths whatever the number of ranks"""
    rng = np.random.default_rng(seed)
    mu = np.log(mean_ops) - 0.5 * sigma * sigma
    return np.maximum(1, rng.lognormal(mu, sigma, size=n_rec).astype(np.int64))


def make_paf_batch_torch(seed, n_rec, mean_ops, pool_bytes, device, use_m=False, sigma=0.5,
                         neg_frac=0.5, n_ops=None):
    """Same mixture as make_paf_batch, generated in HBM.  Returns a dict of torch tensors
    (ops int32 view of the packed u32 ops, op_off/src offsets int64) plus host n_ops.
    n_ops: the records' op counts (default: record_lengths(seed, ...))."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    if n_ops is None:
        n_ops = record_lengths(seed, n_rec, mean_ops, sigma)
    else:
        n_ops = np.asarray(n_ops, dtype=np.int64)
        n_rec = len(n_ops)
    op_off_h = np.zeros(n_rec + 1, dtype=np.int64)
    np.cumsum(n_ops, out=op_off_h[1:])
    total = int(op_off_h[-1])
    op_off = torch.from_numpy(op_off_h).to(device)
    rec_id = torch.repeat_interleave(torch.arange(n_rec, device=device), torch.from_numpy(n_ops).to(device))
    idx_in_rec = torch.arange(total, device=device) - op_off[rec_id]
    is_match = (idx_in_rec & 1) == 0
    del idx_in_rec
    u = torch.rand(total, device=device, generator=g)
    code = torch.where(u < 0.6, OP_X, torch.where(u < 0.8, OP_I, OP_D)).to(torch.int32)
    ln = torch.empty(total, device=device).geometric_(1.0 / 3.0, generator=g).to(torch.int32)
    heavy = torch.rand(total, device=device, generator=g) < 0.01
    hv = torch.randint(50, 2001, (total,), device=device, generator=g, dtype=torch.int32)
    ln = torch.where(heavy, hv, ln)
    del heavy, hv, u
    ln = torch.where(code == OP_X, torch.ones_like(ln), ln)
    if use_m:
        code = torch.where(code == OP_X, torch.full_like(code, OP_M), code)
    ml = torch.empty(total, device=device).geometric_(1.0 / 24.0, generator=g).to(torch.int32)
    code = torch.where(is_match, torch.full_like(code, OP_M if use_m else OP_EQ), code)
    ln = torch.where(is_match, ml, ln)
    del ml, is_match
    ops = (ln << 4) | code

    def rec_sum(mask):
        c = torch.zeros(total + 1, dtype=torch.int64, device=device)
        torch.cumsum(torch.where(mask, ln, torch.zeros_like(ln)).to(torch.int64), 0, out=c[1:])
        return c[op_off[1:]] - c[op_off[:-1]]
    mx = rec_sum((code == OP_M) | (code == OP_EQ) | (code == OP_X))
    si = rec_sum(code == OP_I)
    sd = rec_sum(code == OP_D)
    del code, ln, rec_id
    t_len, q_len = mx + sd, mx + si
    need = int(max(int(t_len.max()), int(q_len.max()))) + 64
    pool_bytes = max(int(pool_bytes), need)

    def pool():
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
        p = lut[torch.randint(0, 4, (pool_bytes,), device=device, generator=g)]
        k = max(1, pool_bytes // 1000)
        p[torch.randint(0, pool_bytes, (k,), device=device, generator=g)] = ord("N")
        run = 200
        starts = torch.randint(0, max(1, pool_bytes - run), (max(1, pool_bytes // (20 * run)),),
                               device=device, generator=g)
        idx = (starts[:, None] + torch.arange(run, device=device)[None, :]).reshape(-1)
        p[idx] |= 0x20
        return p
    t_pool, q_pool = pool(), pool()
    t_off = (torch.rand(n_rec, device=device, generator=g, dtype=torch.float64)
             * (pool_bytes - t_len).to(torch.float64)).to(torch.int64)
    q_off = (torch.rand(n_rec, device=device, generator=g, dtype=torch.float64)
             * (pool_bytes - q_len).to(torch.float64)).to(torch.int64)
    strand = (torch.rand(n_rec, device=device, generator=g) < neg_frac).to(torch.uint8)
    return dict(ops=ops.contiguous(), op_off=op_off, strand_neg=strand, t_pool=t_pool,
                q_pool=q_pool, t_src_off=t_off, t_src_len=t_len, q_src_off=q_off, q_src_len=q_len,
                n=n_rec, n_ops=total, mx=mx, i=si, d=sd)


def torch_batch_record_to_numpy(tb, i):
    """one record of a torch batch as the numpy dict make_paf_batch returns (n = 1)"""
    a, b = int(tb["op_off"][i]), int(tb["op_off"][i + 1])
    ops = tb["ops"][a:b].cpu().numpy().view(np.uint32)
    to, tl = int(tb["t_src_off"][i]), int(tb["t_src_len"][i])
    qo, ql = int(tb["q_src_off"][i]), int(tb["q_src_len"][i])
    return dict(ops=ops, op_off=np.array([0, b - a], dtype=np.uint64),
                strand_neg=np.array([int(tb["strand_neg"][i])], dtype=np.uint8),
                t_pool=tb["t_pool"][to:to + tl].cpu().numpy(), q_pool=tb["q_pool"][qo:qo + ql].cpu().numpy(),
                t_src_off=np.zeros(1, np.uint64), t_src_len=np.array([tl], dtype=np.uint64),
                q_src_off=np.zeros(1, np.uint64), q_src_len=np.array([ql], dtype=np.uint64))


def paf_text_torch(tb, t_name=b"tchr", q_name=b"qchr", mapq=60):
    """the PAF file of a torch batch (make_paf_batch_torch), built in HBM: one line per record,
    `<q>\\t<qlen>\\t<qs>\\t<qe>\\t<strand>\\t<t>\\t<tlen>\\t<ts>\\t<te>\\t0\\t0\\t<mapq>\\tcg:Z:<cigar>\\n` -> uint8 tensor.
    The 12 columns in front of the CIGAR are formatted on the host (one short string per record), the CIGAR text —
    all of the bytes — on the device: digits per op, an exclusive scan, one scatter per digit position."""
    import torch
    dev = tb["ops"].device
    n, n_ops = tb["n"], tb["n_ops"]
    ops = tb["ops"].view(torch.int32)
    ln = (ops >> 4).to(torch.int64) & 0x0FFFFFFF
    code = (ops & 15).to(torch.int64)
    nd = torch.ones(n_ops, dtype=torch.int64, device=dev)
    p = 10
    for _ in range(9):
        nd += (ln >= p).to(torch.int64)
        p *= 10
    tlen = nd + 1                                          # digits + the op letter
    cig_end = torch.cumsum(tlen, 0)                        # inclusive: end of every op's text inside the CIGAR stream
    op_off = tb["op_off"]
    rec_cig = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rec_cig[1:] = cig_end[op_off[1:] - 1]                  # every record has >= 1 op
    # host: the columns in front of the CIGAR
    qs, ql = tb["q_src_off"].cpu().numpy(), tb["q_src_len"].cpu().numpy()
    ts, tl = tb["t_src_off"].cpu().numpy(), tb["t_src_len"].cpu().numpy()
    neg = tb["strand_neg"].cpu().numpy()
    qn, tn = int(tb["q_pool"].numel()), int(tb["t_pool"].numel())
    heads = [b"%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t0\t0\t%d\tcg:Z:" % (
        q_name, qn, qs[i], qs[i] + ql[i], b"-" if neg[i] else b"+", t_name, tn, ts[i], ts[i] + tl[i], mapq)
        for i in range(n)]
    hl = np.fromiter((len(h) for h in heads), dtype=np.int64, count=n)
    head_all = torch.from_numpy(np.frombuffer(b"".join(heads), dtype=np.uint8).copy()).to(dev)
    hl_d = torch.from_numpy(hl).to(dev)
    # line i = head i + cigar i + '\n'
    line_len = hl_d + (rec_cig[1:] - rec_cig[:-1]) + 1
    line_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(line_len, 0, out=line_off[1:])
    total = int(line_off[-1])
    out = torch.empty(total, dtype=torch.uint8, device=dev)
    # heads
    hoff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(hl_d, 0, out=hoff[1:])
    hrec = torch.repeat_interleave(torch.arange(n, device=dev), hl_d)
    hpos = line_off[hrec] + (torch.arange(int(hoff[-1]), device=dev) - hoff[hrec])
    out[hpos] = head_all
    out[line_off[1:] - 1] = ord("\n")
    del hrec, hpos
    # cigar text: op k's text ends at  line_off[r] + hl[r] + (cig_end[k] - rec_cig[r])
    rec = torch.repeat_interleave(torch.arange(n, device=dev), op_off[1:] - op_off[:-1])
    end = cig_end + (line_off[:-1] + hl_d - rec_cig[:-1])[rec]
    del rec, cig_end
    letters = torch.tensor(list(OP_CHARS.encode() if isinstance(OP_CHARS, str) else OP_CHARS), dtype=torch.uint8, device=dev)
    out[end - 1] = letters[code.clamp(max=len(letters) - 1)]
    v = ln.clone()
    for d in range(10):
        m = nd > d
        if not bool(m.any()):
            break
        out[(end - 2 - d)[m]] = (v[m] % 10 + 48).to(torch.uint8)
        v //= 10
    return out
