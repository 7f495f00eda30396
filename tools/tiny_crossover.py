#!/usr/bin/env python3
"""Where the one-wave-per-alignment kernels (kernels_tiny.inc) stop paying: host-buffer scans of n reads (PBC096, both ends)
with every batch on them (TINY_MAX_ENDS=4096) against the throughput kernels (NO_TINY=1); ms per call, best of 30 after warm-up."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from qcat_amd import native, scanner
det = scanner.factory(kit=sys.argv[1] if len(sys.argv) > 1 else "PBC096")
kit = native.NativeKit(det.descriptor(ends=native.ENDS_BOTH))
ctx = native.NativeContext(0)
for n in (1, 4, 16, 32, 64, 128, 256, 512, 1024, 2048):
    reads = synth.synth_batch(n, 5, det.layouts, 1, 0, error_rate=0.08)
    b, o = native.pack_reads(reads)
    row = []
    for tiny in (True, False):
        native.set_option("TINY_MAX_ENDS", 4096 if tiny else None)
        native.set_option("NO_TINY", None if tiny else 1)
        for _ in range(5):
            ref = ctx.scan(kit, b, o)
        best = 1e9
        for _ in range(30):
            t = time.perf_counter(); got = ctx.scan(kit, b, o); best = min(best, time.perf_counter() - t)
        row.append((best * 1e3, got.tobytes()))
    assert row[0][1] == row[1][1]
    print("%5d reads: tiny %.3f ms, throughput kernels %.3f ms" % (n, row[0][0], row[1][0]))

# ---- qcat_scan_sequences (scan() of whole sequences): the same kernels against the general kernel (one lane per sequence)
print("scan() of whole sequences (read interiors, ~460 letters):")
for n in (1, 16, 256, 4096, 30000, 100000):
    reads = synth.synth_batch(min(n, 2000), 6, det.layouts, 1, 0, error_rate=0.08)
    seqs = [r[150:-150] for r in reads]
    seqs = (seqs * (n // len(seqs) + 1))[:n]
    b, o = native.pack_reads(seqs)
    row = []
    for waves in (True, False):
        native.set_option("NO_TINY", None if waves else 1)
        ref = ctx.scan_sequences(kit, b, o)
        best = 1e9
        for _ in range(3 if n > 1000 else 10):
            t = time.perf_counter(); got = ctx.scan_sequences(kit, b, o); best = min(best, time.perf_counter() - t)
        row.append((best * 1e3, got.tobytes()))
    native.set_option("NO_TINY", None)
    assert row[0][1] == row[1][1]
    print("%6d sequences: one wave per alignment %.3f ms, general kernel %.3f ms" % (n, row[0][0], row[1][0]))
