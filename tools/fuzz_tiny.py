#!/usr/bin/env python3
"""Parity sweep of the one-wave-per-alignment kernels on the GPU box (kernels_tiny.inc): random kit selections (every
shipped kit, kit auto = all templates, the dual scanner), ends, error rates and configurations (scores of the adapter
matrix, linear gap, barcode context length, extracted-barcode extension, max_align_length), small batches with truncated,
odd-lettered and degenerate reads; records, count vector AND every intermediate (per-template raw score and end, regions,
every per-barcode raw score) against the CPU oracle.
    python tools/fuzz_tiny.py FIRST LAST"""
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

first, last = int(sys.argv[1]), int(sys.argv[2])
lib = native.HipLibrary.get().lib
ctx = native.NativeContext(0)
bad = ran = 0
for seed in range(first, last):
    rng = random.Random(seed)
    mode = rng.choice(["epi2me", "epi2me", "dual"])
    kit = None if mode == "dual" else rng.choice([None] + sorted(scanner.get_kits()))
    det = scanner.factory(mode=mode, kit=kit)
    cfg = config.qcatConfig()
    if rng.random() < 0.5:
        cfg.barcode_context_length = rng.choice([0, 4, 6, 9, 11])
        cfg.extracted_barcode_extension = rng.choice([0, 5, 11, 14])
        cfg.max_align_length = rng.choice([60, 100, 120, 150, 160])
    if rng.random() < 0.3:                                   # other linear gaps (open == extend keeps the batch on the path)
        cfg.gap_open = cfg.gap_extend = rng.choice([1, 2, 3, 5])
    ends = rng.choice([native.ENDS_BOTH, native.ENDS_5P])
    d = det.descriptor(qcat_config=cfg, ends=ends)
    nl = len(det.layouts)
    t5 = rng.randrange(nl)
    t3 = rng.randrange(nl) if ends == native.ENDS_BOTH else -1
    n = rng.choice([1, 2, 3, 9, 30])
    reads = synth.synth_batch(n, seed * 7 + 1, det.layouts, t5, t3, error_rate=rng.choice([0.0, 0.05, 0.1, 0.2, 0.35]))
    for i in range(n):
        k = rng.randrange(8)
        if k == 0:
            reads[i] = reads[i][:rng.randrange(0, 330)]
        elif k == 1:
            p = rng.randrange(0, max(1, len(reads[i])))
            reads[i] = reads[i][:p] + rng.choice(["N", "R", "x", "*", "NNNNNNNN"]) + reads[i][p + 1:]
        elif k == 2:
            reads[i] = reads[i].lower()
        elif k == 3:
            reads[i] = rng.choice(["", "A", "N" * rng.randrange(1, 200), "ACGT" * rng.randrange(1, 90)])
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(d, reads, counts=True, trace=True, rows=True, threads=8)
    nk = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    native.set_option("TINY_MAX_ENDS", 4096)
    recs, traces, rows = ctx.scan(nk, bases, offsets, counts=cnt, trace=True, rows=True)
    on_path = lib.qcat_ctx_tiny_ends(ctx.handle) == n * (2 if ends == native.ENDS_BOTH else 1)
    ok = recs.tobytes() == o_recs.tobytes() and np.array_equal(cnt, o_cnt) and np.array_equal(rows, o_rows) and \
        all(np.array_equal(traces[name], o_traces[name]) for name in native.TRACE_DTYPE.names)
    ran += on_path
    bad += not ok
    print("seed %4d %-7s kit %-14s ends %d n %2d gap %d ctx %2d ext %2d window %3d on the path %d: %s" % (
        seed, mode, kit, 2 if ends == native.ENDS_BOTH else 1, n, cfg.gap_open, cfg.barcode_context_length,
        cfg.extracted_barcode_extension, cfg.max_align_length, on_path, "ok" if ok else "MISMATCH"), flush=True)
print("%d seeds, %d on the one-wave kernels, %d mismatches" % (last - first, ran, bad))
sys.exit(1 if bad or ran != last - first else 0)
