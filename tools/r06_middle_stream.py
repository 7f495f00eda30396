#!/usr/bin/env python3
"""where the file loop's time goes with --detect-middle (whole reads go to the device): the library's own stage clocks and,
with QCAT_HIP_PIPELINE_TRACE=1, the timeline of the segments.   python tools/r06_middle_stream.py [reads]"""
import ctypes as C
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from qcat_amd import config, native, scanner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
hip = native.HipLibrary.get()
lib = hip.lib
ctx = native.NativeContext(0)
tmp = tempfile.mkdtemp(prefix="qcat_mid_")
gen = native.NativeKit(scanner.factory(kit="PBC096").descriptor())
sp = native.SynthParams(seed=9, n_reads=n, insert_len=600, lead_min=5, lead_max=40, error_rate=0.08, no_adapter_fraction=0.05, tpl_5p=1, tpl_3p=0)
b = C.c_void_p()
hip.check(lib.qcat_batch_synthesize(ctx.handle, gen.handle, C.byref(sp), C.byref(b)))
nb, nr = C.c_uint64(), C.c_uint32()
hip.check(lib.qcat_batch_info(b, C.byref(nr), C.byref(nb)))
bases = np.zeros(nb.value, dtype=np.uint8)
offs = np.zeros(n + 1, dtype=np.uint64)
hip.check(lib.qcat_batch_download(ctx.handle, b, bases.ctypes.data, offs.ctypes.data))
lib.qcat_batch_destroy(b)
raw = bases.tobytes()
qual = b"I" * 4096
big = os.path.join(tmp, "big.fastq")
with open(big, "wb", buffering=32 << 20) as fh:
    for i in range(n):
        s = raw[int(offs[i]):int(offs[i + 1])]
        fh.write(b"@read%d runid=bench ch=%d\n" % (i, 1 + i % 512))
        fh.write(s)
        fh.write(b"\n+\n")
        fh.write(qual[:len(s)])
        fh.write(b"\n")
print("file: %.2f GB, %d reads" % (os.path.getsize(big) / 1e9, n))
cfg = config.get_default_config()
for middle in (False, True, "auto"):
    det = scanner.factory(kit=None) if middle == "auto" else scanner.factory(kit="PBC096", scan_middle_adapter=middle)
    kit = det._native_kit(det.layouts, cfg, native.ENDS_BOTH)
    for rep in range(3):
        sink = tempfile.TemporaryFile()
        t0 = time.perf_counter()
        st = native.FastqFile.demux_stream(big, det._context(), kit, det.layouts, False, kit_auto=(middle == "auto"), trim=True, min_read_length=100,
                                           tsv_fd=sink.fileno(), out_fd=None, out_dir=None)[4]
        dt = time.perf_counter() - t0
        sink.close()
        print("detect-middle %s run %d: %.3f s = %.1f M reads/s; busy: parse %.3f scan %.3f write %.3f; %d segments" % (
            middle, rep, dt, n / dt / 1e6, st["parse_s"], st["scan_s"], st["write_s"], st["segments"]), flush=True)
