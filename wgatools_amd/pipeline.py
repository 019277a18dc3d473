"""Batch drivers over device-resident data: the per-batch body of `converter::paf2maf`
(converter.rs:196-263) + `stat_paf` (stat.rs:87-105) expressed as C-ABI calls.

Buffers are torch tensors (torch = allocator / stream plumbing only); all work is done by
libwgahip.so on the torch current stream.
"""
import numpy as np

from . import engine


class Paf2MafStatJob:
    """stat (K1) -> row layout (scan) -> gap insertion (K2) over one resident batch."""

    def __init__(self, eng, tb, with_text=False, out=None):
        """out: a caller-owned output buffer (uint8, at least the rows' bytes + 64), e.g. one reserved at process start."""
        import torch
        self.torch = torch
        self.eng = eng
        self.tb = tb
        dev = tb["ops"].device
        n = tb["n"]
        self.n, self.n_ops = n, tb["n_ops"]
        self.batch = engine.Batch(tb["ops"], tb["op_off"], tb["strand_neg"], n, tb["n_ops"])
        self.counts = torch.zeros((n, 11), dtype=torch.int64, device=dev)
        self.diag = torch.zeros((n, 3), dtype=torch.int64, device=dev)
        self.tile_ws = torch.zeros(eng.lib.wga_tile_ws_bytes(tb["n_ops"]), dtype=torch.uint8, device=dev)
        self.t_row_off = torch.zeros(n, dtype=torch.int64, device=dev)
        self.q_row_off = torch.zeros(n, dtype=torch.int64, device=dev)
        self.rec_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        # output size is known from the generator's class sums (a host driver learns it from
        # rec_off[n] after the layout call)
        rows = int((tb["t_src_len"] + tb["i"]).sum() + (tb["q_src_len"] + tb["d"]).sum())
        self.pre = None
        if with_text:  # room for "a score=..\ns\t..\t" style text around the rows
            self.pre = tuple(torch.full((n,), v, dtype=torch.int32, device=dev) for v in (48, 40, 2))
            rows += n * 90
        self.out_bytes = rows
        if out is not None and out.numel() < rows + 64:
            raise ValueError("output buffer too small: %d < %d" % (out.numel(), rows + 64))
        self.out = out[: rows + 64] if out is not None else torch.empty(rows + 64, dtype=torch.uint8, device=dev)

    def bind_stream(self):
        self.eng.set_stream(self.torch.cuda.current_stream().cuda_stream)

    def stat(self):
        self.eng.cigar_stat(self.batch, self.counts, self.diag, self.tile_ws)

    def layout(self):
        p = self.pre or (None, None, None)
        self.eng.paf2maf_layout(self.n, self.counts, self.tb["t_src_len"], self.tb["q_src_len"],
                                p[0], p[1], p[2], self.t_row_off, self.q_row_off, self.rec_off)

    def expand(self):
        tb = self.tb
        self.eng.paf2maf_expand(self.batch, self.counts, self.tile_ws, tb["t_pool"],
                                tb["t_pool"].numel(), tb["t_src_off"], tb["t_src_len"],
                                tb["q_pool"], tb["q_pool"].numel(), tb["q_src_off"],
                                tb["q_src_len"], self.out, self.t_row_off, self.q_row_off,
                                self.diag)

    def step(self):
        self.stat()
        self.layout()
        self.expand()

    # algorithmic bytes: what the kernels must move at minimum
    def algorithmic_bytes(self):
        tb = self.tb
        t_len = int(tb["t_src_len"].sum())
        q_len = int(tb["q_src_len"].sum())
        L2 = int((tb["mx"] + tb["i"] + tb["d"]).sum()) * 2
        return dict(stat=4 * self.n_ops + 88 * self.n,
                    expand=4 * self.n_ops + t_len + q_len + L2)

    def record_rows(self, i):
        """host copies of record i's two rows (for parity spot checks)"""
        c = self.counts[i].cpu().numpy()
        tl = int(self.tb["t_src_len"][i]) + int(c[3] + c[7])
        ql = int(self.tb["q_src_len"][i]) + int(c[5] + c[9])
        to, qo = int(self.t_row_off[i]), int(self.q_row_off[i])
        return (self.out[to:to + tl].cpu().numpy().tobytes(),
                self.out[qo:qo + ql].cpu().numpy().tobytes())


def output_bytes(tb):
    """bytes of the two gapped rows of every record of a generated batch (+ slack), known from the generator's class sums"""
    return int((tb["t_src_len"] + tb["i"]).sum() + (tb["q_src_len"] + tb["d"]).sum()) + 64
