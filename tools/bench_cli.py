#!/usr/bin/env python3
"""End-to-end rate of the driver (qcat_amd.cli) on a synthetic FASTQ file: the native file loop (qcat_fastq_demux_stream:
segments of the file through read | scan | write) against the Python loop of the same driver (QCAT_AMD_NO_NATIVE_FASTQ=1:
Python parser, batches of 4000, Python writers), with and without --detect-middle / --filter-barcodes.

    python tools/bench_cli.py [reads, default 2000000] [python-loop reads, default 100000]

Prints one JSON line: reads/s of every variant, the native path's split (parse / scan / write seconds and the parse rate in
GB/s), and that the outputs of the two paths are byte-identical (sha256 per file) on the first `python-loop reads` reads."""
import ctypes as C
import hashlib
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from qcat_amd import cli, config, native, scanner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
n_py = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
det = scanner.factory(kit="PBC096")
hip = native.HipLibrary.get()
lib = hip.lib
kit = native.NativeKit(det.descriptor())
ctx = native.NativeContext(0)
tmp = tempfile.mkdtemp(prefix="qcat_cli_bench_", dir=os.environ.get("QCAT_BENCH_TMP", None))


def write_fastq(path, count):
    sp = native.SynthParams(seed=9, n_reads=count, insert_len=600, lead_min=5, lead_max=40, error_rate=0.08,
                            no_adapter_fraction=0.05, tpl_5p=1, tpl_3p=0)
    b = C.c_void_p()
    hip.check(lib.qcat_batch_synthesize(ctx.handle, kit.handle, C.byref(sp), C.byref(b)))
    nb, nr = C.c_uint64(), C.c_uint32()
    hip.check(lib.qcat_batch_info(b, C.byref(nr), C.byref(nb)))
    bases = np.zeros(nb.value, dtype=np.uint8)
    offs = np.zeros(count + 1, dtype=np.uint64)
    hip.check(lib.qcat_batch_download(ctx.handle, b, bases.ctypes.data, offs.ctypes.data))
    lib.qcat_batch_destroy(b)
    raw = bases.tobytes()
    qual = b"I" * 4096
    # (a 32 MB buffer: the file reaches the page cache in large pieces, as a basecaller or a copy writes it -- a file that
    #  arrives in 8 KB writes sits in single pages and every mapping operation on it costs several times as much)
    with open(path, "wb", buffering=32 << 20) as fh:
        for i in range(count):
            s = raw[int(offs[i]):int(offs[i + 1])]
            fh.write(b"@read%d runid=bench ch=%d\n" % (i, 1 + i % 512))
            fh.write(s)
            fh.write(b"\n+\n")
            fh.write(qual[:len(s)])
            fh.write(b"\n")


def run(fq, kitname, out, tsv, native_path, tsv_file=None, middle=False, filt=False):
    """one run of the driver; the TSV goes to a StringIO (compared by the caller) or, at size, to a file like the shell's `> calls.tsv`"""
    if native_path:
        os.environ.pop("QCAT_AMD_NO_NATIVE_FASTQ", None)
    else:
        os.environ["QCAT_AMD_NO_NATIVE_FASTQ"] = "1"
    buf = open(tsv_file, "w") if tsv_file else io.StringIO()
    t0 = time.perf_counter()
    dist = cli.qcat_cli(reads_fq=fq, kit=kitname, mode="epi2me", nobatch=False, out=out, min_qual=None, tsv=tsv,
                        output=None if (out or tsv) else os.path.join(tmp, "stream.fastq"), threads=1, trim=True, adapter_yaml=None,
                        quiet=True, filter_barcodes=filt, middle_adapter=middle, min_read_length=100,
                        qcat_config=config.get_default_config(), tsv_stream=buf)
    dt = time.perf_counter() - t0
    if tsv_file:
        buf.close()
        return dt, dist, None
    return dt, dist, buf.getvalue()


def sha_dir(d):
    out = {}
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            out[f] = hashlib.sha256(fh.read()).hexdigest()
    return out


res = {"reads": n, "python_loop_reads": n_py, "host_threads": int(os.environ.get("QCAT_HOST_THREADS", "0")) or None}
small = os.path.join(tmp, "small.fastq")
write_fastq(small, n_py)
# ---- identical outputs of the two paths (per-barcode files with trimming, and TSV), fixed kit and kit auto -------------
same = True
for kitname in ("PBC096", "auto"):
    dt_py, dist_py, _ = run(small, kitname, os.path.join(tmp, "py_" + kitname), False, False)
    dt_nat, dist_nat, _ = run(small, kitname, os.path.join(tmp, "nat_" + kitname), False, True)
    same = same and sha_dir(os.path.join(tmp, "py_" + kitname)) == sha_dir(os.path.join(tmp, "nat_" + kitname)) and dist_py == dist_nat
    _, _, tsv_py = run(small, kitname, None, True, False)
    _, _, tsv_nat = run(small, kitname, None, True, True)
    same = same and tsv_py == tsv_nat
    res["kit_%s_%d_reads" % (kitname, n_py)] = {"python_loop_reads_per_s": round(n_py / dt_py, 1), "native_reads_per_s": round(n_py / dt_nat, 1)}
res["outputs_identical"] = bool(same)
# ---- the same with the driver's two per-read / per-batch flags (round 5: both on the native loop) ----------------------
same_flags = True
for label, kw in (("detect_middle", {"middle": True}), ("filter_barcodes", {"filt": True}), ("both", {"middle": True, "filt": True})):
    dt_py, dist_py, tsv_py = run(small, "PBC096", None, True, False, **kw)
    dt_nat, dist_nat, tsv_nat = run(small, "PBC096", None, True, True, **kw)
    same_flags = same_flags and tsv_py == tsv_nat and dist_py == dist_nat
    res["flags_%s_%d_reads" % (label, n_py)] = {"python_loop_reads_per_s": round(n_py / dt_py, 1), "native_reads_per_s": round(n_py / dt_nat, 1)}
res["outputs_identical_with_flags"] = bool(same_flags)
# ---- the native path at size: TSV (calls only) and per-barcode FASTQ files (the whole file is written again) ----------
big = os.path.join(tmp, "big.fastq")
write_fastq(big, n)
size = os.path.getsize(big)
f = native.FastqFile(big)           # (warm the page cache the way a just-written file is)
f.close()
best = None
for _ in range(3):
    t0 = time.perf_counter()
    f = native.FastqFile(big)
    dt = time.perf_counter() - t0
    f.close()
    best = dt if best is None else min(best, dt)
res["ingest"] = {"file_gb": round(size / 1e9, 3), "open_s": round(best, 4), "parse_gb_per_s": round(size / best / 1e9, 2),
                 "reads_per_s": round(n / best, 1)}
for label, out, tsv in (("tsv", None, True), ("per_barcode_fastq", os.path.join(tmp, "big_out"), False)):
    dt = min(run(big, "PBC096", out, tsv, True, tsv_file=os.path.join(tmp, "big.tsv") if tsv else None)[0] for _ in range(2))
    res["native_" + label] = {"reads_per_s": round(n / dt, 1), "seconds": round(dt, 3)}
    # the split of one more run, from the library's own clock (busy seconds per stage; the stages run side by side)
    sink = tempfile.TemporaryFile()
    if out and not os.path.exists(out):
        os.makedirs(out)
    st = native.FastqFile.demux_stream(big, ctx, kit, det.layouts, False, kit_auto=False, trim=True, min_read_length=100,
                                       tsv_fd=sink.fileno() if tsv else None, out_fd=None, out_dir=out)[4]
    sink.close()
    res["native_" + label]["split_s"] = {k: round(st[k], 4) for k in ("parse_s", "scan_s", "write_s", "total_s")}
    res["native_" + label]["segments"] = st["segments"]
for label, kw in (("detect_middle", {"middle": True}), ("filter_barcodes", {"filt": True})):
    dt = min(run(big, "PBC096", None, True, True, tsv_file=os.path.join(tmp, "big_flags.tsv"), **kw)[0] for _ in range(2))
    res["native_tsv_" + label] = {"reads_per_s": round(n / dt, 1), "seconds": round(dt, 3)}
# ---- the driver's DEFAULT: kit auto, one vote per batch of 4000 reads (qcat/cli.py:500) -- every batch a call of its own;
#      round 4: chunks of 64 batches per call (qcat_scan_batches_auto_ptrs: one vote per batch on the device), reads as pointers into the mapping
for label, chunk in (("one_call_per_batch", "1"), ("default", None)):          # (round 3's loop; chunks of 64 batches per call)
    native.set_option("AUTO_CHUNK", int(chunk) if chunk else None)
    dt = min(run(big, "auto", None, True, True, tsv_file=os.path.join(tmp, "big_auto.tsv"))[0] for _ in range(2))
    res["native_tsv_kit_auto_" + label] = {"reads_per_s": round(n / dt, 1), "seconds": round(dt, 3)}
res["note"] = ("host-bound: segments of the file are split on the host threads, the scan reads heads and tails of the reads in place "
               "(whole reads with --detect-middle), the writer formats beside the scan (total_s = the demux call, seconds = the whole "
               "driver run incl. context set-up); per-barcode FASTQ output rewrites every byte of the input")
print(json.dumps(res))
