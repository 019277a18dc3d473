"""TEST HARNESS (round 6: it lived in the package as wgatools_amd/dist_cli.py until the multi-GPU drivers were reduced to one).
One process per GPU over a PAF file, the collectives of wgatools_amd/multigpu.py (torch.distributed: "nccl" = RCCL on the GPU
box, gloo on the emulator build) around the single-GPU calls of the C-ABI: it checks that the partition rule and the
collectives bench.py --gpus N measures give the bytes of the product's own multi-GPU driver, `wgatools --gpus N`
(north_star: "records shard naturally by hash(target_name) across the 8 GPUs of one node with an RCCL
reduce over xGMI only for the global stat / pafcov totals").

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tests/dist_cli.py paf2maf  in.paf -g target.fa -q query.fa -o out.maf
        tests/dist_cli.py pafcov   in.paf -o out.bed [--spread]
        tests/dist_cli.py totals   in.paf                      (the 11 global stat counters, one JSON line)

Every rank reads the whole PAF (host parsing is the same work on every rank; the kernels are not), keeps the records
`fnv1a64(target_name) % N` gives it and runs the single-GPU calls of the C-ABI on them (`wgatools_amd.engine`); the bytes
are those of `wgatools paf2maf` / `wgatools pafcov` on one GPU:

  * paf2maf  — converter.rs:176-265 writes in INPUT ORDER.  A record's output size is known after K1 + the layout scan;
    one all-reduce of the global size vector gives every rank the file offsets of its records and it `pwrite`s them in
    place (`multigpu.ordered_offsets`, `write_ordered`): no row byte crosses GPUs.
  * pafcov   — pafcov.rs:18-64.  With target sharding a target's coverage array lives on one rank: no collective, the
    ranks write their targets' BED text at offsets from one all-reduce of the per-target text sizes.  `--spread` deals
    the records out round robin instead (one hot target): every rank accumulates its share of every target and the
    partial coverages are summed with one reduce-scatter per target (`multigpu.hot_target_coverage`), each rank
    formatting the slice it ends up with.
  * totals   — all-reduce of the 11 counters (88 bytes).

Scope: a reference driver over the C-ABI, not a second command line — clean `cg:Z:` PAF (no csv quoting, no `cs` tags),
plain or gzip FASTA read by every rank; a rank works through its records in resident batches of `--chunk-bytes` of CIGAR text
(paf2maf: sizes first, rows second).  Errors of the hot path (invalid op, invalid
base, the insert_str panic) end the run on every rank with the reference's message for the first failing record in input
order; the output keeps the records in front of it, as the reference's reader loop does.
`--lib PATH` binds another build of the library (the CPU tests pass the SIMT-emulator build and run over gloo); without it
the in-tree libwgahip.so is loaded and the run fails loudly when no MI355X is visible.
"""
import argparse
import gzip
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgatools_amd import _lib, engine, multigpu  # noqa: E402

NONE = 0xFFFFFFFFFFFFFFFF


class HotPathError(Exception):
    pass


class RecordError(HotPathError):
    """a hot-path error of one record (input index): the ranks end the run together on the smallest index"""

    def __init__(self, index, msg):
        HotPathError.__init__(self, msg)
        self.index = int(index)


# ---- host side: the same text rules as wga_host.cpp ---------------------------------------------------------------
def read_text(path):
    with open(path, "rb") as f:
        head = f.read(2)
    if head == b"\x1f\x8b":
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def parse_paf(text):
    """records of a clean PAF text: csv crate framing without quoting (paf.rs:24-30) — lines end at \\n, \\r\\n; empty
    lines and lines starting with '#' are skipped; tab-separated; twelve typed fields, then tags"""
    recs = []
    for ln, line in enumerate(text.split(b"\n")):
        if line.endswith(b"\r"):
            line = line[:-1]
        if not line or line.startswith(b"#"):
            continue
        f = line.split(b"\t")
        if len(f) < 12:
            raise HotPathError("CSV deserialize error: record %d: missing field" % len(recs))
        cg = None
        for tag in f[12:]:
            if tag.startswith(b"cg:Z:"):
                cg = tag[5:]
                break
        if cg is None:
            raise HotPathError("CIGAR start tag not found")
        recs.append(dict(qname=f[0].decode(), qlen=int(f[1]), qstart=int(f[2]), qend=int(f[3]), neg=f[4] == b"-",
                         tname=f[5].decode(), tlen=int(f[6]), tstart=int(f[7]), tend=int(f[8]), mapq=int(f[11]),
                         cigar=cg))
    return recs


class Fasta:
    """name -> (pool offset, length); the pool = all sequences, line ends stripped, case kept (wga_host.cpp Faidx::load:
    names end at the first white space, a repeated name keeps its first sequence)"""

    def __init__(self, path):
        text = read_text(path)
        self.contigs = {}
        parts = []
        size = 0
        cur, cur_off = None, 0

        def close():
            if cur is not None and cur not in self.contigs:
                self.contigs[cur] = (cur_off, size - cur_off)
        for line in text.split(b"\n"):
            if line.endswith(b"\r"):
                line = line[:-1]
            if line.startswith(b">"):
                close()
                cur = line[1:].split(None, 1)[0].decode() if len(line) > 1 and line[1:].split(None, 1) else ""
                cur_off = size
            elif cur is not None:
                parts.append(line)
                size += len(line)
        close()
        self.pool = np.frombuffer(b"".join(parts), dtype=np.uint8)

    def fetch(self, name, beg, end_incl):
        """faidx_fetch_seq64 clipping (Faidx::fetch): -> (pool offset, length)"""
        if name not in self.contigs:
            raise HotPathError("HTS library error by sequence `%s` not found in the FASTA index" % name)
        off, L = self.contigs[name]
        if end_incl < beg:
            beg = end_incl
        beg = 0 if beg < 0 else (L if L <= beg else beg)
        end = 0 if end_incl < 0 else (L - 1 if L <= end_incl else end_incl)
        return off + beg, max(0, end + 1 - beg)


def cigar_op_token_at(cigar, op_idx):
    """the op char of packed op `op_idx` of a CIGAR text (a length >= 2^28 is several packed ops)"""
    k, num = 0, 0
    for ch in cigar.decode("latin-1"):
        if ch.isdigit():
            num = num * 10 + int(ch)
            continue
        pieces = max(1, (num + engine.OP_MAX_LEN - 1) // engine.OP_MAX_LEN) if ch in "ID" else 1
        if op_idx < k + pieces:
            return ch
        k += pieces
        num = 0
    return "?"


# ---- the ranks ------------------------------------------------------------------------------------------------------
class Ranks:
    def __init__(self, lib_path):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        on_gpu = lib_path is None
        if self.world > 1 or os.environ.get("WGA_DIST_FORCE"):   # WGA_DIST_FORCE=1: the collectives at world size 1 too
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29534")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group("nccl" if on_gpu else "gloo")
            self.dist = dist
        if on_gpu:
            torch.cuda.set_device(local)       # fails loudly without a GPU: there is no CPU path
        self.on_gpu = on_gpu
        self.dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
        self.eng = engine.Engine(local if on_gpu else 0, _lib.load(lib_path) if lib_path else None)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def torch_done(self):
        """torch's stream and the library's are different streams: what torch enqueued (zeroing a tensor the kernels will
        write) must have happened before the library's kernels start"""
        if self.on_gpu:
            self.torch.cuda.synchronize()

    def all_min(self, value):
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device=self.dev)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def all_sum(self, value):
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device=self.dev)
        if self.dist is not None:
            self.dist.all_reduce(t)
        return int(t.item())

    def close(self):
        self.eng.close()
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def pack_records(eng, recs, idx):
    """packed CSR batch of the records idx (host packer of the C-ABI); a tokeniser error raises RecordError with the
    record's input index: only the rank that owns the record sees it, so the callers agree on the first failing index
    (Ranks.all_min) before any other collective"""
    ops, off = [], [0]
    for i in idx:
        o, err, (eo, el) = eng.pack_cigar(recs[i]["cigar"])
        if err:
            tok = recs[i]["cigar"][eo:eo + el].decode("latin-1")
            raise RecordError(i, ("Parse `%s` Into Integer Error" % tok) if err == 3 else ("CIGAR OP `%s` invalid" % tok))
        ops.append(o)
        off.append(off[-1] + len(o))
    ops = np.concatenate(ops) if ops else np.zeros(0, np.uint32)
    strand = np.array([1 if recs[i]["neg"] else 0 for i in idx], dtype=np.uint8)
    return eng.make_batch(ops, np.array(off, dtype=np.uint64), strand)


def chunks_of(recs, idx, max_text):
    """idx cut into runs whose CIGAR text stays below max_text bytes (one resident batch each)"""
    out, cur, size = [], [], 0
    for i in idx:
        ln = len(recs[int(i)]["cigar"])
        if cur and size + ln > max_text:
            out.append(cur)
            cur, size = [], 0
        cur.append(int(i))
        size += ln
    if cur:
        out.append(cur)
    return out


def maf_piece(eng, recs, idx, tf, qf, d_tp, d_qp, want_rows):
    """One resident batch of paf2maf: the records idx (input order).  Returns (sizes, rec_off, rows or None, err) where
    err = None or RecordError of the first failing record of the piece — the arrays then cover the records in front of it
    (converter.rs:209-235: slices are fetched, then the query is reverse-complemented, then the CIGAR is walked)."""
    err = None
    t_off, t_len, q_off, q_len, pre_t, pre_q, blob, blob_off, kept, packed = [], [], [], [], [], [], [], [0], [], []
    for i in idx:
        r = recs[i]
        try:
            to, tl = tf.fetch(r["tname"], r["tstart"], r["tend"] - 1)
            qo, ql = qf.fetch(r["qname"], r["qstart"], r["qend"] - 1)
            o, e, (eo, el) = eng.pack_cigar(r["cigar"])
            if e:
                tok = r["cigar"][eo:eo + el].decode("latin-1")
                raise HotPathError(("Parse `%s` Into Integer Error" % tok) if e == 3 else ("CIGAR OP `%s` invalid" % tok))
        except HotPathError as ex:
            err = RecordError(i, str(ex))
            break
        kept.append(i)
        packed.append(o)
        t_off.append(to); t_len.append(tl); q_off.append(qo); q_len.append(ql)
        a = "a score=%d\ns\t%s\t%d\t%d\t+\t%d\t" % (r["mapq"], r["tname"], r["tstart"], r["tend"] - r["tstart"], r["tlen"])
        q = "\ns\t%s\t%d\t%d\t%s\t%d\t" % (r["qname"], r["qlen"] - r["qend"] if r["neg"] else r["qstart"],
                                            r["qend"] - r["qstart"], "-" if r["neg"] else "+", r["qlen"])
        for t in (a.encode(), q.encode(), b"\n\n"):
            blob.append(t)
            blob_off.append(blob_off[-1] + len(t))
        pre_t.append(len(a)); pre_q.append(len(q))
    m = len(kept)
    if m == 0:
        return np.zeros(0, np.int64), np.zeros(1, np.uint64), None, err
    op_off = np.zeros(m + 1, dtype=np.uint64)
    np.cumsum([len(o) for o in packed], out=op_off[1:])
    batch = eng.make_batch(np.concatenate(packed), op_off, np.array([1 if recs[i]["neg"] else 0 for i in kept], dtype=np.uint8))
    up = lambda v, dt: eng.upload(np.asarray(v, dtype=dt))
    counts, diag, tws = eng.cigar_stat(batch, want_tiles=want_rows)
    d_tl, d_ql = up(t_len, np.uint64), up(q_len, np.uint64)
    tro, qro, reco = eng.paf2maf_layout(m, counts, d_tl, d_ql, up(pre_t, np.uint32), up(pre_q, np.uint32), up([2] * m, np.uint32))
    rec_off = reco.numpy()
    sizes = (rec_off[1:] - rec_off[:-1]).astype(np.int64)
    if not want_rows:
        return sizes, rec_off, None, err
    out = eng.empty(int(rec_off[-1]) + 64, np.uint8)
    eng.paf2maf_expand(batch, counts, tws, d_tp, len(tf.pool), up(t_off, np.uint64), d_tl, d_qp, len(qf.pool),
                       up(q_off, np.uint64), d_ql, out, tro, qro, diag)
    c, tr = counts.numpy(), tro.numpy()
    dst = np.empty(3 * m, dtype=np.uint64)
    dst[0::3] = rec_off[:-1]
    dst[1::3] = tr + np.asarray(t_len, dtype=np.uint64) + c["ins_bp"] + c["inv_ins_bp"]
    dst[2::3] = rec_off[1:] - 2
    eng.scatter_bytes(3 * m, eng.upload(np.frombuffer(b"".join(blob), dtype=np.uint8)), up(blob_off, np.uint64), out, up(dst, np.uint64))
    dg = diag.numpy()
    for k in range(m):
        g = dg[k]
        if int(g["bad_base_pos"]) == NONE and int(g["bad_op_idx"]) == NONE and int(g["panic_op_idx"]) == NONE:
            continue
        if int(g["bad_base_pos"]) != NONE:          # utils.rs:97: reverse_complement runs before the CIGAR is walked
            msg = "Invalid Base: `%s`" % chr(int(qf.pool[q_off[k] + q_len[k] - 1 - int(g["bad_base_pos"])]))
        elif int(g["bad_op_idx"]) < int(g["panic_op_idx"]):
            msg = "CIGAR OP `%s` invalid" % cigar_op_token_at(recs[kept[k]]["cigar"], int(g["bad_op_idx"]))
        else:
            msg = "panic: String::insert_str beyond the end of the fetched sequence (cigar.rs:507,513)"
        return sizes[:k], rec_off[:k + 1], out.numpy(), RecordError(kept[k], msg)
    return sizes, rec_off, out.numpy(), err


def run_paf2maf(R, args):
    eng = R.eng
    recs = parse_paf(read_text(args.input))
    tf, qf = Fasta(args.target), Fasta(args.query)
    n = len(recs)
    owner = multigpu.owners([r["tname"] for r in recs], R.world)
    mine = [int(i) for i in multigpu.my_records(owner, R.rank)]
    header = ("#maf version=1.6 convert_from=paf t_seq_path=%s q_seq_path=%s\n" % (args.target, args.query)).encode()
    d_tp, d_qp = eng.upload(tf.pool), eng.upload(qf.pool)
    first_err = None
    # pass 1: the size of every record's text (K1 + the layout scan; no row byte yet), one resident batch at a time
    sizes = {}
    for piece in chunks_of(recs, mine, args.chunk_bytes):
        sz, _, _, err = maf_piece(eng, recs, piece, tf, qf, d_tp, d_qp, want_rows=False)
        for i, v in zip(piece, sz):
            sizes[i] = int(v)
        if err is not None:
            first_err = err
            break
    if first_err is not None:
        mine = [i for i in mine if i < first_err.index]
    my_sizes = np.array([sizes[i] for i in mine], dtype=np.int64)
    glob, total = multigpu.ordered_offsets(n, np.array(mine, dtype=np.int64), my_sizes, R.dist, R.dev)
    glob = dict(zip(mine, (glob + len(header)).tolist()))
    total += len(header)
    if R.rank == 0:
        multigpu.write_ordered(args.output, np.frombuffer(header, dtype=np.uint8), [0], [len(header)], [0], total=total, create=True)
    R.barrier()
    # pass 2: the rows, written where they belong
    for piece in chunks_of(recs, mine, args.chunk_bytes):
        sz, rec_off, rows, err = maf_piece(eng, recs, piece, tf, qf, d_tp, d_qp, want_rows=True)
        k = len(sz)
        if k:
            multigpu.write_ordered(args.output, rows, rec_off[:-1][:k], sz, [glob[i] for i in piece[:k]])
        if err is not None:
            first_err = err if first_err is None or err.index < first_err.index else first_err
            break
    # every rank learns the first failing record (input order): the file ends in front of it
    bad_at = R.all_min(first_err.index if first_err is not None else n)
    if bad_at < n:
        keep_bytes = R.all_sum(sum(sizes[i] for i in mine if i < bad_at)) + len(header)
        R.barrier()
        if R.rank == 0:
            os.truncate(args.output, keep_bytes)
        R.barrier()
        if first_err is not None and first_err.index == bad_at:
            sys.stderr.write("ERROR %s\n" % first_err)
        return 1
    R.barrier()
    return 0


def end_together(R, first_err, n):
    """every rank learns the smallest failing record index (one MIN all-reduce, the FIRST collective after the per-rank
    work); the owner of that record prints the reference's message; True = the run ends with status 1 on every rank"""
    bad_at = R.all_min(first_err.index if first_err is not None else n)
    if bad_at >= n:
        return False
    if first_err is not None and first_err.index == bad_at:
        sys.stderr.write("ERROR %s\n" % first_err)
    return True


def paf_targets(recs):
    """names in first-appearance order, array length = target_length of the first record seen (cmd_pafcov)"""
    names, tid, length = [], {}, []
    for r in recs:
        if r["tname"] not in tid:
            tid[r["tname"]] = len(names)
            names.append(r["tname"])
            length.append(r["tlen"])
    return names, tid, length


def run_pafcov(R, args):
    torch, eng = R.torch, R.eng
    recs = parse_paf(read_text(args.input))
    names, tid, length = paf_targets(recs)
    nt = len(names)
    t_owner = multigpu.owners(names, R.world)
    if args.spread:
        mine = np.arange(R.rank, len(recs), R.world)                # records dealt out round robin
        my_targets = list(range(nt))                                # every rank holds (its share of) every target
    else:
        mine = np.array([i for i, r in enumerate(recs) if t_owner[tid[r["tname"]]] == R.rank], dtype=np.int64)
        my_targets = [t for t in range(nt) if t_owner[t] == R.rank]
    local_id = {t: k for k, t in enumerate(my_targets)}
    cov_len = np.array([length[t] for t in my_targets], dtype=np.uint64)
    cov_off = np.zeros(len(my_targets), dtype=np.uint64)
    total = 0
    for k in range(len(my_targets)):
        cov_off[k] = total
        total += (int(cov_len[k]) + 3) & ~3
    cov = torch.zeros(total + 4, dtype=torch.int32, device=R.dev)
    R.torch_done()
    first_err = None
    if len(mine) and len(my_targets):
        d_off, d_len = eng.upload(cov_off), eng.upload(cov_len)
        try:
            for piece in chunks_of(recs, mine, args.chunk_bytes):      # one resident batch at a time into the same arrays
                batch = pack_records(eng, recs, piece)
                target_id = np.array([local_id[tid[recs[i]["tname"]]] for i in piece], dtype=np.uint32)
                t_start = np.array([recs[i]["tstart"] for i in piece], dtype=np.uint64)
                eng.pafcov_accumulate(batch, eng.upload(target_id), eng.upload(t_start), d_off, d_len, cov, total)
                eng.sync()
            eng.pafcov_finalize(len(my_targets), d_off, d_len, cov)
            eng.sync()
        except RecordError as ex:    # only this rank knows: carry on to the agreement below, with nothing to write
            first_err = ex
    # the first failing record in input order ends the run on EVERY rank, before any other collective (pafcov is a
    # buffered driver: nothing is written, pafcov.rs:29-53)
    if end_together(R, first_err, len(recs)):
        return 1
    # pieces of BED text: (target, first position, int32 coverage tensor of the positions)
    pieces = []
    for t in range(nt):
        if args.spread:
            k = local_id[t]
            part = cov[int(cov_off[k]): int(cov_off[k]) + int(cov_len[k])]
            lo, hi, sl = multigpu.hot_target_coverage(part, R.dist)  # reduce-scatter: I end up with positions [lo, hi)
            pieces.append((t, R.rank, lo, sl[: hi - lo].contiguous()))
        elif t_owner[t] == R.rank:
            k = local_id[t]
            pieces.append((t, 0, 0, cov[int(cov_off[k]): int(cov_off[k]) + int(cov_len[k])]))
    R.torch_done()            # the collectives ran on torch's stream; K9 below runs on the library's
    # format my pieces on the device (K9), learn everybody's text sizes, write in (target, slice) order
    texts, sizes = [], np.zeros(nt * R.world, dtype=np.int64)
    for t, slot, p0, part in pieces:
        cnt = int(part.numel())
        if cnt == 0:
            texts.append((t, slot, np.zeros(0, np.uint8)))
            continue
        name = eng.upload(np.frombuffer(names[t].encode(), dtype=np.uint8))
        loff = eng.pafcov_format(name, part, p0, cnt)
        nbytes = int(loff.numpy()[-1])
        txt = eng.empty(nbytes + 16, np.uint8)
        eng.pafcov_format(name, part, p0, cnt, line_off=loff, out=txt)
        texts.append((t, slot, txt.numpy()[:nbytes]))
        sizes[t * R.world + slot] = nbytes
    slots = np.array([t * R.world + slot for t, slot, _ in texts], dtype=np.int64)
    glob, total_bytes = multigpu.ordered_offsets(nt * R.world, slots, sizes[slots] if len(slots) else np.zeros(0, np.int64), R.dist, R.dev)
    if R.rank == 0:
        multigpu.write_ordered(args.output, np.zeros(0, np.uint8), [], [], [], total=total_bytes, create=True)
    R.barrier()
    for (t, slot, b), g in zip(texts, glob):
        if len(b):
            multigpu.write_ordered(args.output, b, [0], [len(b)], [int(g)])
    R.barrier()
    return 0


def run_totals(R, args):
    torch, eng = R.torch, R.eng
    recs = parse_paf(read_text(args.input))
    owner = multigpu.owners([r["tname"] for r in recs], R.world)
    mine = multigpu.my_records(owner, R.rank)
    tot = torch.zeros(11, dtype=torch.int64, device=R.dev)
    R.torch_done()
    part = torch.zeros(11, dtype=torch.int64, device=R.dev)
    R.torch_done()
    first_err = None
    try:
        for piece in chunks_of(recs, [int(i) for i in mine], args.chunk_bytes):
            batch = pack_records(eng, recs, piece)
            counts, diag, _ = eng.cigar_stat(batch, want_tiles=False)
            bad = diag.numpy()["bad_op_idx"]
            if (bad != np.uint64(NONE)).any():     # parse_paf_to_cigar rejects ops outside M = X I D (cigar.rs:629-707)
                k = int(np.flatnonzero(bad != np.uint64(NONE))[0])
                raise RecordError(piece[k], "CIGAR OP `%s` invalid" % cigar_op_token_at(recs[piece[k]]["cigar"], int(bad[k])))
            eng.counts_total(len(piece), counts, part)
            eng.sync()
            tot += part
            R.torch_done()      # the add runs on torch's stream: it must have read `part` before the next piece overwrites it
    except RecordError as ex:
        first_err = ex
    if end_together(R, first_err, len(recs)):
        return 1
    multigpu.allreduce_totals(tot, R.dist)
    if R.rank == 0:
        print(json.dumps(dict(zip(engine.COUNTS_DTYPE.names, [int(x) for x in tot.cpu().tolist()]), records=len(recs), ranks=R.world)))
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python tests/dist_cli.py", description=__doc__.split("\n\n")[0])
    ap.add_argument("--lib", default=None, help="library to bind instead of the in-tree libwgahip.so (tests: the emulator build; runs over gloo)")
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("paf2maf")
    p.add_argument("input")
    p.add_argument("-g", "--target", required=True)
    p.add_argument("-q", "--query", required=True)
    p.add_argument("-o", "--output", required=True)
    p.add_argument("--chunk-bytes", type=int, default=64 << 20, help="CIGAR text per resident batch of a rank (default 64 MiB)")
    p = sub.add_parser("pafcov")
    p.add_argument("input")
    p.add_argument("-o", "--output", required=True)
    p.add_argument("--spread", action="store_true", help="deal the records out round robin and sum the partial coverages with a reduce-scatter per target")
    p.add_argument("--chunk-bytes", type=int, default=64 << 20)
    p = sub.add_parser("totals")
    p.add_argument("input")
    p.add_argument("--chunk-bytes", type=int, default=64 << 20)
    args = ap.parse_args(argv)
    out = getattr(args, "output", None)
    if out and out.endswith((".gz", ".bz2", ".xz")):
        # every rank writes its records at their offsets in the PLAIN output; a compressed stream has no such offsets.  The
        # single-process command line deflates `.gz` outputs on the device (wga_bgzf_compress).
        sys.stderr.write("ERROR IO error:the multi-rank writer places records at their plain offsets: `%s` would not be a "
                         "compressed file; write a plain file here, or a .gz with the wgatools command line\n" % out)
        return 1
    R = Ranks(args.lib)
    try:
        rc = {"paf2maf": run_paf2maf, "pafcov": run_pafcov, "totals": run_totals}[args.cmd](R, args)
    except HotPathError as e:
        sys.stderr.write("ERROR %s\n" % e)
        rc = 1
    R.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
