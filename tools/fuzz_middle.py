#!/usr/bin/env python3
"""Parity sweep of the bit-sliced INTERIOR adapter scan of --detect-middle on the GPU box (csrc/kernels_abs_mid.inc): random kit
selections (named kits, kit auto, the dual kit), chimeric reads (the read joined to itself / to its reverse complement), inserts of
random length so that one big tile of 2048 interiors holds many length classes (front padding of hundreds of rows), N runs and
lower case inside interiors, reads without an interior; both kernel forms (one wave per tile / the two-wave pipeline), sometimes
a plane buffer that is too small; a few reads with interiors beyond the packed path (the one-wave kernels).  Records and count vector against the CPU oracle, and the tiles that ran bit-sliced are counted
through qcat_ctx_middle_bitslice_tiles.
    python tools/fuzz_middle.py FIRST LAST"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                   # noqa: E402
import oracle_lib                    # noqa: E402
import synth                         # noqa: E402
from qcat_amd import native, scanner   # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
lib = native.HipLibrary.get().lib
comp = {"A": "T", "T": "A", "G": "C", "C": "G"}
bad = on_path = 0
for seed in range(first, last):
    rng = random.Random(seed)
    mode = rng.choice(["epi2me", "epi2me", "epi2me", "dual"])
    kit = None if mode == "dual" else rng.choice([None] + sorted(scanner.get_kits()))
    det = scanner.factory(mode=mode, kit=kit, scan_middle_adapter=True)
    nl = len(det.layouts)
    t5 = rng.randrange(nl)
    t3 = rng.choice([-1, rng.randrange(nl)])
    n = rng.choice([300, 700, 1500])
    base = synth.synth_batch(n, seed * 11 + 3, det.layouts, t5, t3, error_rate=rng.choice([0.0, 0.05, 0.1]))
    reads = []
    for j, r in enumerate(base):
        k = rng.randrange(8)
        if k == 0:
            reads.append(r + r)
        elif k == 1:
            rc = "".join(comp.get(ch, "N") for ch in reversed(r))
            reads.append(r[:len(r) // 2] + rc + r[len(r) // 2:])
        elif k == 2:
            reads.append(r[:150 + rng.randrange(0, 500)] + r[-170:])
        elif k == 3:
            reads.append(r + "".join(rng.choice("ACGT") for _ in range(rng.randrange(50, 1800))) + r)
        elif k == 4:
            p = rng.randrange(150, max(151, len(r) - 150))
            reads.append(r[:p] + "N" * rng.randrange(1, 40) + r[p:])
        elif k == 5:
            reads.append(r.lower() if rng.random() < 0.5 else r[:rng.randrange(0, 420)])
        elif k == 6 and j % 16 == 6:
            # an interior beyond the packed interior scan's 16 384 letters (the one-wave kernels: k_midw_*), with or without the
            # read / its reverse complement / a stretch of N inside
            inner = rng.choice(["", r, "".join(comp.get(ch, "N") for ch in reversed(r)), "N" * 50])
            filler = "".join(rng.choice("ACGT") for _ in range(rng.randrange(8200, 9500)))
            reads.append(r[:200] + filler + inner + filler[::-1] + r[-200:])
        else:
            reads.append(r)
    d = det.descriptor()
    want, want_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    nk = native.NativeKit(d)
    ctx = native.NativeContext(0)
    bases, offsets = native.pack_reads(reads)
    one_wave = rng.choice(["1", "0"])
    rows = rng.choice([None, None, str(rng.randrange(200, 6000))])
    native.set_option("MIDDLE_ABS_MIN", 1)
    native.set_option("MIDDLE_ABS_ONE_WAVE", int(one_wave))
    native.set_option("MIDDLE_ABS_ROWS", int(rows) if rows else None)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    got = ctx.scan(nk, bases, offsets, counts=cnt)
    tiles = (C.c_uint32 * 4)()
    lib.qcat_ctx_middle_bitslice_tiles(ctx.handle, tiles)
    on_path += tiles[0] > 0
    long_reads = lib.qcat_ctx_middle_wave_reads(ctx.handle)
    mism = int(np.count_nonzero(got != want))
    ok = mism == 0 and np.array_equal(cnt, want_cnt)
    print("seed %3d %-6s %-16s t5 %2d t3 %2d n %4d one-wave %s rows %-5s big tiles %d / %d, tiles of 128 left to binary16 %d / %d, long interiors on waves %d, 997: %d : %s" % (
        seed, mode, kit, t5, t3, n, one_wave, rows, tiles[0], tiles[1], tiles[2], tiles[3], long_reads, int((got["exit_status"] == 997).sum()),
        "ok" if ok else "MISMATCH %d" % mism), flush=True)
    bad += not ok
print("seeds %d..%d: %d failures, the bit-sliced interior scan ran in %d" % (first, last - 1, bad, on_path))
sys.exit(1 if bad else 0)
