"""what bounds the copy-out of a large result: the D2H copy engine (pinned, piece by piece) or the page-cache writes"""
import os, sys, time, threading
import numpy as np, torch
dev = torch.device("cuda", 0)
N = 4 << 30
src = torch.empty(N, dtype=torch.uint8, device=dev); src.fill_(65)
for piece_mb in (16, 64, 256):
    P = piece_mb << 20
    bufs = [torch.empty(P, dtype=torch.uint8).pin_memory() for _ in range(2)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N // P):
        bufs[i & 1].copy_(src[i * P:(i + 1) * P], non_blocking=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("D2H pinned, %3d MB pieces, sync each: %.1f GB/s" % (piece_mb, N / dt / 1e9), flush=True)
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N // P):
        bufs[i & 1].copy_(src[i * P:(i + 1) * P], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("D2H pinned, %3d MB pieces, queued:    %.1f GB/s" % (piece_mb, N / dt / 1e9), flush=True)
host = np.full(16 << 20, 66, dtype=np.uint8)
for nthr in (1, 4, 8, 16):
    path = "/tmp/wga_wr_test.bin"
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    total = 8 << 30
    npieces = total // host.nbytes
    def work(t):
        for p in range(t, npieces, nthr): os.pwrite(fd, host, p * host.nbytes)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t0
    os.close(fd); os.remove(path)
    print("pwrite into /tmp, %2d threads, 16 MB pieces: %.1f GB/s" % (nthr, total / dt / 1e9), flush=True)
