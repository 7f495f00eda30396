#!/usr/bin/env python3
"""The kit-auto file loop alone (qcat_fastq_demux with kit_auto, TSV to /dev/null) on a warm context, by worker count:
    python tools/bench_auto_file.py [reads, default 1000000]
prints one JSON line: seconds of the demux call (the library's own clock) for QCAT_HIP_AUTO_WORKERS = 1, 2, 4, first and
second call of each."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402
from qcat_amd import config, native, scanner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
det1 = scanner.factory(kit="PBC096")
hip = native.HipLibrary.get()
lib = hip.lib
kit1 = native.NativeKit(det1.descriptor())
ctx = native.NativeContext(0)
sp = native.SynthParams(seed=9, n_reads=n, insert_len=600, lead_min=5, lead_max=40, error_rate=0.08, no_adapter_fraction=0.05, tpl_5p=1, tpl_3p=0)
b = C.c_void_p()
hip.check(lib.qcat_batch_synthesize(ctx.handle, kit1.handle, C.byref(sp), C.byref(b)))
nb, nr = C.c_uint64(), C.c_uint32()
hip.check(lib.qcat_batch_info(b, C.byref(nr), C.byref(nb)))
bases = np.zeros(nb.value, dtype=np.uint8)
offs = np.zeros(n + 1, dtype=np.uint64)
hip.check(lib.qcat_batch_download(ctx.handle, b, bases.ctypes.data, offs.ctypes.data))
lib.qcat_batch_destroy(b)
raw = bases.tobytes()
tmp = tempfile.mkdtemp(prefix="qcat_auto_", dir=os.environ.get("QCAT_BENCH_TMP", None))
path = os.path.join(tmp, "reads.fastq")
qual = b"I" * 4096
with open(path, "wb") as fh:
    for i in range(n):
        s = raw[int(offs[i]):int(offs[i + 1])]
        fh.write(b"@read%d ch=%d\n" % (i, 1 + i % 512) + s + b"\n+\n" + qual[:len(s)] + b"\n")
det = scanner.factory()                                       # kit auto
kit = det._native_kit(det.layouts, config.qcatConfig(), native.ENDS_BOTH)
res = {"reads": n}
sink = open(os.devnull, "wb")
for workers, chunk in ((1, 1), (4, 1), (1, 16), (1, 64), (2, 64), (4, 64), (4, 256)):
    native.set_option("AUTO_WORKERS", workers)
    native.set_option("AUTO_CHUNK", chunk)
    c = native.NativeContext(0)                               # a fresh context: the first call pays the helpers' set-up
    runs = []
    for rep in range(3):
        fq = native.FastqFile(path)
        recs, skipped, st = fq.demux(c, kit, det.layouts, False, batch_size=4000, kit_auto=True, trim=True, min_read_length=100, tsv_fd=sink.fileno())
        fq.close()
        runs.append(round(st["total_s"], 4))
    res["workers_%d_chunk_%d" % (workers, chunk)] = {"total_s": runs, "reads_per_s_warm": round(n / min(runs[1:]), 1), "replays": int(lib.qcat_ctx_graph_replays(c.handle))}
    del c
print(json.dumps(res))
