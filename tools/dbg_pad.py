#!/usr/bin/env python3
"""debug: small PBC096 batch on the padded super-tiles against the oracle (argv: reads, seed)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, synth
from qcat_amd import native, scanner
n = int(sys.argv[1]); seed = int(sys.argv[2])
det = scanner.factory(mode="epi2me", kit="PBC096")
reads = synth.synth_batch(n, seed, det.layouts, 1, 0, error_rate=0.08)
d = det.descriptor(ends=native.ENDS_BOTH)
want = oracle_lib.scan(d, reads, threads=8)
bases, offsets = native.pack_reads(reads)
ctx = native.NativeContext(0)
got = ctx.scan(native.NativeKit(d), bases, offsets)
bad = np.nonzero(got != want)[0]
print("reads", n, "mismatches", len(bad), "first", bad[:12].tolist())
for i in bad[:4]:
    print(i, got[i], want[i])
