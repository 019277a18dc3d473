"""how fast can 8 GB leave pinned-like host memory for a NEW file under /tmp (page cache)?  pwrite by N threads (what DevStreamer
does), the same after fallocate / ftruncate, and threads copying into a shared mapping of the file"""
import os, sys, time, threading, mmap, subprocess
import numpy as np
print(subprocess.run("df -T /tmp | tail -1; grep -E ' /tmp | / ' /proc/mounts | head -3", shell=True, stdout=subprocess.PIPE).stdout.decode().strip())
total = 8 << 30
piece = 16 << 20
host = np.full(piece, 66, dtype=np.uint8)
path = "/tmp/wga_wr_test.bin"
npieces = total // piece
def timed(label, prep, work, nthr):
    if os.path.exists(path): os.remove(path)
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    t0 = time.perf_counter()
    ctx = prep(fd)
    th = [threading.Thread(target=work, args=(fd, ctx, t, nthr)) for t in range(nthr)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t0
    if ctx is not None:
        del ctx
    os.close(fd)
    print("%-46s %2d threads: %.1f GB/s" % (label, nthr, total / dt / 1e9), flush=True)
def pw(fd, ctx, t, nthr):
    for p in range(t, npieces, nthr): os.pwrite(fd, host, p * piece)
def pw_blocks(fd, ctx, t, nthr):          # every thread its own contiguous eighth
    a, z = npieces * t // nthr, npieces * (t + 1) // nthr
    for p in range(a, z): os.pwrite(fd, host, p * piece)
def mm(fd, m, t, nthr):
    for p in range(t, npieces, nthr): m[p * piece:(p + 1) * piece] = host
def prep_none(fd): return None
def prep_falloc(fd): os.posix_fallocate(fd, 0, total); return None
def prep_trunc(fd): os.ftruncate(fd, total); return None
def prep_map(fd):
    os.ftruncate(fd, total)
    return np.frombuffer(mmap.mmap(fd, total, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE), dtype=np.uint8)
for nthr in (8, 16, 32):
    timed("pwrite, pieces dealt round robin", prep_none, pw, nthr)
timed("pwrite, a contiguous share per thread", prep_none, pw_blocks, 8)
timed("pwrite after ftruncate to the final size", prep_trunc, pw, 8)
timed("pwrite after posix_fallocate", prep_falloc, pw, 8)
for nthr in (8, 16, 32):
    timed("copy into a shared mapping (after ftruncate)", prep_map, mm, nthr)
os.remove(path)
