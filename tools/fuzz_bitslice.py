#!/usr/bin/env python3
"""Parity sweep of the bit-sliced barcode kernels on the GPU box (kernels_bitslice.inc): random kit selections, ends,
error rates and configurations (barcode context length, extracted-barcode extension, max_align_length -- which change
the number of shared columns, the direction, the own columns and the row counts), batches with truncated and odd reads,
records and count vector against the CPU oracle.  Configurations away from the defaults have no built-in generated
kernels, so those kits are compiled at run time (hipRTC) first.
    QCAT_HIP_BITSLICE_MIN=2048 python tools/fuzz_bitslice.py FIRST LAST"""
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
from qcat_amd import config, native, scanner   # noqa: E402

if native.get_option("BITSLICE_MIN") is None:
    native.set_option("BITSLICE_MIN", 2048)
first, last = int(sys.argv[1]), int(sys.argv[2])
lib = native.HipLibrary.get().lib
bad = ran_bs = 0
for seed in range(first, last):
    rng = random.Random(seed)
    mode = rng.choice(["epi2me", "epi2me", "dual"])
    kit = None if mode == "dual" else rng.choice([None] + sorted(scanner.get_kits()))
    det = scanner.factory(mode=mode, kit=kit)
    cfg = config.qcatConfig()
    custom = rng.random() < 0.5
    if custom:
        cfg.barcode_context_length = rng.choice([4, 6, 8, 9, 11])
        cfg.extracted_barcode_extension = rng.choice([5, 8, 11, 14])
        cfg.max_align_length = rng.choice([100, 120, 150])
    ends = rng.choice([native.ENDS_BOTH, native.ENDS_5P])
    d = det.descriptor(qcat_config=cfg, ends=ends)
    nl = len(det.layouts)
    t5 = rng.randrange(nl)
    t3 = rng.randrange(nl) if ends == native.ENDS_BOTH else -1
    n = rng.choice([5000, 9000])
    reads = synth.synth_batch(n, seed * 7 + 1, det.layouts, t5, t3, error_rate=rng.choice([0.0, 0.05, 0.1, 0.2]))
    for i in range(0, n, 13):
        k = (i // 13) % 4
        if k == 0:
            reads[i] = reads[i][:rng.randrange(0, 260)]
        elif k == 1:
            reads[i] = reads[i][:50] + "N" + reads[i][51:]
        elif k == 2:
            reads[i] = reads[i].lower()
    want, want_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    nk = native.NativeKit(d, jit=True)
    ctx = native.NativeContext(0)
    native.HipLibrary.get().check(lib.qcat_ctx_set_timing(ctx.handle, 1))
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    got = ctx.scan(nk, bases, offsets, counts=cnt)
    names = (C.c_char_p * 16)()
    ms = (C.c_float * 16)()
    ran = [names[i].decode() for i in range(lib.qcat_ctx_last_timing(ctx.handle, names, ms, 16))]
    bs = "k_barcode_bitslice" in ran
    ran_bs += bs
    mism = int(np.count_nonzero(got != want))
    ok = mism == 0 and np.array_equal(cnt, want_cnt)
    print("seed %3d %-6s %-16s ends %d custom %d ctx %2d ext %2d L %3d n %5d bit-sliced %d info %x : %s" % (
        seed, mode, kit, ends, custom, cfg.barcode_context_length, cfg.extracted_barcode_extension, cfg.max_align_length, n, bs,
        nk.describe()["bitslice_groups"], "ok" if ok else "MISMATCH %d" % mism), flush=True)
    bad += not ok
print("seeds %d..%d: %d failures, bit-sliced kernels ran in %d" % (first, last - 1, bad, ran_bs))
sys.exit(1 if bad else 0)
