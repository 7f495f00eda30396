#!/usr/bin/env python3
"""--detect-middle on a batch that holds a few very long reads: the interiors beyond the packed interior scan (16 384 letters) on
the one-wave kernels (k_midw_*, default) against the general kernel, one lane per read (NO_TINY=1); ms per resident scan."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from qcat_amd import native, scanner
rng = random.Random(3)
det = scanner.factory(kit="NBD103/NBD104", scan_middle_adapter=True)
kit = native.NativeKit(det.descriptor(ends=native.ENDS_BOTH, scan_middle=True))
ctx = native.NativeContext(0)
lib = native.HipLibrary.get().lib
base = synth.synth_batch(20000, 5, det.layouts, 1, 0, error_rate=0.08)
for n_long, length in ((0, 0), (1, 50000), (10, 50000), (100, 30000), (1000, 20000)):
    reads = list(base)
    for i in range(n_long):
        r = base[i]
        reads[i] = r[:220] + "".join(rng.choice("ACGT") for _ in range(length)) + r[-220:]
    b, o = native.pack_reads(reads)
    row = []
    for waves in (True, False):
        native.set_option("NO_TINY", None if waves else 1)
        ref = ctx.scan(kit, b, o)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); got = ctx.scan(kit, b, o); best = min(best, time.perf_counter() - t)
        row.append((best * 1e3, got.tobytes(), lib.qcat_ctx_middle_wave_reads(ctx.handle)))
    native.set_option("NO_TINY", None)
    assert row[0][1] == row[1][1]
    print("20000 reads, %4d of them %5d letters long: %8.2f ms with the one-wave kernels (%d reads on them), %8.2f ms with the general kernel" % (
        n_long, length + 440, row[0][0], row[0][2], row[1][0]))
