#!/usr/bin/env python3
"""End-to-end rate of the Python driver (qcat_amd.cli) on a synthetic FASTQ: parse -> batches of 4000
-> native scan -> trimming -> per-barcode FASTQ files.  Bound by Python I/O, not by the GPU."""
import os, sys, time, tempfile, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from qcat_amd import cli, config, native, scanner

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
det = scanner.factory(kit="PBC096")
hip = native.HipLibrary.get(); lib = hip.lib
kit = native.NativeKit(det.descriptor()); ctx = native.NativeContext(0)
sp = native.SynthParams(seed=9, n_reads=n, insert_len=600, lead_min=5, lead_max=40, error_rate=0.08,
                        no_adapter_fraction=0.05, tpl_5p=1, tpl_3p=0)
b = C.c_void_p(); hip.check(lib.qcat_batch_synthesize(ctx.handle, kit.handle, C.byref(sp), C.byref(b)))
nb = C.c_uint64(); nr = C.c_uint32(); hip.check(lib.qcat_batch_info(b, C.byref(nr), C.byref(nb)))
bases = np.zeros(nb.value, dtype=np.uint8); offs = np.zeros(n + 1, dtype=np.uint64)
hip.check(lib.qcat_batch_download(ctx.handle, b, bases.ctypes.data, offs.ctypes.data))
tmp = tempfile.mkdtemp(prefix="qcat_cli_bench_")
fq = os.path.join(tmp, "reads.fastq")
raw = bases.tobytes()
with open(fq, "w") as fh:
    for i in range(n):
        s = raw[int(offs[i]):int(offs[i + 1])].decode()
        fh.write("@read%d ch=1\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
for kitname in ("PBC096", "auto"):
    t0 = time.perf_counter()
    dist = cli.qcat_cli(reads_fq=fq, kit=kitname, mode="epi2me", nobatch=False, out=os.path.join(tmp, "out_" + kitname), min_qual=None,
                        tsv=False, output=None, threads=1, trim=True, adapter_yaml=None, quiet=True, filter_barcodes=False,
                        middle_adapter=False, min_read_length=100, qcat_config=config.get_default_config())
    dt = time.perf_counter() - t0
    print("cli kit=%s: %d reads in %.2f s = %.0f reads/s (%.0f MB FASTQ); barcodes called: %d" %
          (kitname, n, dt, n / dt, os.path.getsize(fq) / 1e6, sum(v for k, v in dist[0].items() if k != "none")))
