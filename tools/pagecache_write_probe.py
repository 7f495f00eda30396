#!/usr/bin/env python3
"""How fast can this box take file writes (page cache)?  The per-barcode FASTQ output of the driver re-emits every byte of
the input: 1.5 GB per million reads.  T threads write `total` bytes as pieces of `piece` bytes into F files (pwrite at
explicit offsets, pieces of one file from different threads), from an anonymous buffer and from a mapped file.
usage: python tools/pagecache_write_probe.py [dir]"""
import mmap
import os
import sys
import tempfile
import threading
import time

d = tempfile.mkdtemp(prefix="pcw_", dir=sys.argv[1] if len(sys.argv) > 1 else None)
print("dir", d, "fs:", os.popen("df -T %s | tail -1" % d).read().strip())
total = 1536 << 20
src_path = os.path.join(d, "src.bin")
with open(src_path, "wb") as fh:
    blk = os.urandom(1 << 20)
    for _ in range(total >> 20):
        fh.write(blk)
fd_src = os.open(src_path, os.O_RDONLY)
mm = mmap.mmap(fd_src, total, prot=mmap.PROT_READ)
anon = bytearray(total)
mv_anon, mv_map = memoryview(anon), memoryview(mm)


def run(label, T, F, piece, src):
    fds = [os.open(os.path.join(d, "out_%d.bin" % i), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666) for i in range(F)]
    n_pieces = total // piece
    def work(t):
        for p in range(t, n_pieces, T):
            f = p % F
            off = (p // F) * piece
            os.pwrite(fds[f], src[p * piece:(p + 1) * piece], off)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for fd in fds: os.close(fd)
    dt_close = time.perf_counter() - t1
    print("%-46s %2d threads %3d files %7d KB pieces: %.3f s  %.1f GB/s; close() of the files %.1f ms" % (label, T, F, piece >> 10, dt, total / dt / 1e9, dt_close * 1e3))
    t2 = time.perf_counter()
    for i in range(F): os.unlink(os.path.join(d, "out_%d.bin" % i))
    print("    unlink of the files %.1f ms" % ((time.perf_counter() - t2) * 1e3))


for T in (1, 4, 16):
    run("anonymous buffer", T, 16, 1 << 20, mv_anon)
run("anonymous buffer", 16, 97, 256 << 10, mv_anon)
run("mapped file", 16, 97, 256 << 10, mv_map)
run("mapped file", 16, 16, 1 << 20, mv_map)
run("mapped file", 16, 1, 1 << 20, mv_map)
run("anonymous buffer, rewrite of existing pages", 16, 16, 1 << 20, mv_anon)
# files that exist already (a second run into the same directory): emptied by O_TRUNC, rewritten, closed
for i in range(97):
    with open(os.path.join(d, "out_%d.bin" % i), "wb") as fh:
        fh.write(blk * 16)
run("anonymous buffer, files existed (O_TRUNC)", 16, 97, 256 << 10, mv_anon)
