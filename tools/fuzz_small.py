#!/usr/bin/env python3
"""Parity sweep of the THROUGHPUT kernels on small batches on the GPU box (round 6: the merged launches k_adapter_multi /
k_barcode_multi, static_generated.inc; the LDS-staged k_adapter_finish / k_pick_kit): random kit selections (every shipped
kit, kit auto = the twelve auto-detect templates, the dual scanner), ends, error rates, batches of 65 .. 6000 reads with truncated,
odd-lettered and degenerate reads, default configuration (the generated static-letter kernels); records and count vector
against the CPU oracle -- with the traces (per-template raw score and end, every per-barcode raw score) on every third
seed -- and, for kit auto, the one-pass call (vote on the device) against the oracle of the voted kit.
    python tools/fuzz_small.py FIRST LAST"""
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
native.set_option("NO_TINY", 1)
bad = 0
for seed in range(first, last):
    rng = random.Random(seed)
    mode = rng.choice(["epi2me", "epi2me", "epi2me", "dual"])
    kit = None if mode == "dual" else rng.choice([None, None] + sorted(scanner.get_kits()))
    det = scanner.factory(mode=mode, kit=kit)
    cfg = config.qcatConfig()
    ends = rng.choice([native.ENDS_BOTH, native.ENDS_BOTH, native.ENDS_5P])
    d = det.descriptor(qcat_config=cfg, ends=ends)
    nl = len(det.layouts)
    t5 = rng.randrange(nl)
    t3 = rng.randrange(nl) if ends == native.ENDS_BOTH else -1
    n = rng.choice([65, 129, 500, 1000, 2500, 4000, 6000])
    reads = synth.synth_batch(n, seed * 7 + 1, det.layouts, t5, t3, error_rate=rng.choice([0.0, 0.05, 0.1, 0.2]),
                              no_adapter_fraction=rng.choice([0.0, 0.05, 0.3]))
    for i in range(0, n, rng.choice([3, 11, 50])):
        k = rng.randrange(6)
        if k == 0:
            reads[i] = reads[i][:rng.randrange(0, 330)]
        elif k == 1:
            p = rng.randrange(0, max(1, len(reads[i])))
            reads[i] = reads[i][:p] + rng.choice(["N", "R", "x", "*", "NNNNNNNN"]) + reads[i][p + 1:]
        elif k == 2:
            reads[i] = reads[i].lower()
        elif k == 3:
            reads[i] = rng.choice(["", "A", "N" * rng.randrange(1, 200), "ACGT" * rng.randrange(1, 90)])
    with_traces = seed % 3 == 0
    nk = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    ctx = native.NativeContext(0)
    if with_traces:
        o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(d, reads, counts=True, trace=True, rows=True, threads=16)
        recs, traces, rows = ctx.scan(nk, bases, offsets, counts=cnt, trace=True, rows=True)
        ok = recs.tobytes() == o_recs.tobytes() and np.array_equal(cnt, o_cnt) and np.array_equal(rows, o_rows) and \
            all(np.array_equal(traces[name], o_traces[name]) for name in native.TRACE_DTYPE.names)
    else:
        o_recs, o_cnt = oracle_lib.scan(d, reads, counts=True, threads=16)
        recs = ctx.scan(nk, bases, offsets, counts=cnt)
        ok = recs.tobytes() == o_recs.tobytes() and np.array_equal(cnt, o_cnt)
        recs2 = ctx.scan(nk, bases, offsets)                       # (a call of the same shape: the captured graph)
        ok = ok and recs2.tobytes() == o_recs.tobytes()
    auto = ""
    if kit is None and mode == "epi2me" and ends == native.ENDS_BOTH:
        got = ctx.scan_auto(nk, bases, offsets)
        if got is None:
            auto = " one-pass: not taken"
            ok = False
        else:
            a_recs, slot = got
            sub = det.get_adapters(d.kit_names[slot])
            o = oracle_lib.scan(det.descriptor(layouts=sub, qcat_config=cfg), reads, threads=16)
            full_index = np.array([det.layouts.index(l) for l in sub] + [-1])
            same = np.array_equal(full_index[o["adapter_idx"]], a_recs["adapter_idx"]) and all(
                np.array_equal(o[name], a_recs[name]) for name in ("barcode_idx", "barcode2_idx", "exit_status", "adapter_end", "trim5p",
                                                                     "trim3p", "raw_score", "score_den"))
            auto = " one-pass (kit %s): %s" % (d.kit_names[slot], "ok" if same else "MISMATCH")
            ok = ok and same
    bad += not ok
    print("seed %4d %-7s kit %-14s ends %d n %4d traces %d:%s %s" % (seed, mode, kit, 2 if ends == native.ENDS_BOTH else 1, n,
                                                                   with_traces, auto, "ok" if ok else "MISMATCH"), flush=True)
print("%d seeds, %d mismatches" % (last - first, bad))
sys.exit(1 if bad else 0)
