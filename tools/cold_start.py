#!/usr/bin/env python3
"""Where the time before a file's first segment goes, in a fresh process (round 6): python tools/cold_start.py [reads]
Stages: interpreter + imports, the library (dlopen of libqcat_hip.so), the HIP runtime + context (hipInit, stream, first
allocations), the scanner + kit (host-side preparation), the first demux call of a small file (kit upload, code objects of the
kernels it launches, graph capture), the second call of the same file."""
import os
import sys
import tempfile
import time

t_start = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
marks = []


def mark(name):
    marks.append((name, time.perf_counter()))


import numpy as np  # noqa: E402,F401
mark("numpy")
from qcat_amd import native, scanner  # noqa: E402
mark("import qcat_amd")
hip = native.HipLibrary.get()
mark("dlopen libqcat_hip.so")
n_dev = hip.lib.qcat_device_count()
mark("hipInit (device count)")
ctx = native.NativeContext(0)
mark("context (stream, first allocations)")
det = scanner.factory(kit="PBC096")
mark("scanner.factory (kit YAML / json)")
kit = native.NativeKit(det.descriptor())
mark("NativeKit (host preparation)")
import synth  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reads = synth.synth_batch(min(n, 2000), 11, det.layouts, 1, 0, error_rate=0.08)
tmp = tempfile.mkdtemp(prefix="qcat_cold_")
path = os.path.join(tmp, "r.fastq")
with open(path, "w") as fh:
    for i in range(n):
        r = reads[i % len(reads)]
        fh.write("@read%d some comment\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
mark("(test file written)")
for rep in range(3):
    with open(os.path.join(tmp, "out%d.tsv" % rep), "wb") as sink:
        st = native.FastqFile.demux_stream(path, ctx, kit, det.layouts, False, kit_auto=False, trim=False, min_read_length=0,
                                           tsv_fd=sink.fileno(), out_fd=None, out_dir=None)[4]
    mark("demux call %d of %d reads (parse %.1f scan %.1f write %.1f ms busy)" % (rep + 1, n, st["parse_s"] * 1e3, st["scan_s"] * 1e3, st["write_s"] * 1e3))
prev = t_start
for name, t in marks:
    print("%8.1f ms  %s" % ((t - prev) * 1e3, name))
    prev = t
print("%8.1f ms  total since interpreter start-up of this script" % ((marks[-1][1] - t_start) * 1e3))
