#!/usr/bin/env python3
"""bench.py -- reads/s of the barcode-demultiplexing hot path on N MI355X GPUs of one node.

    python bench.py                                  # N = 1: BASELINE config 3 (10 M reads, PBC096, 5'+3')
    python bench.py --gpus 8 --steps 20 --warmup 3   # starts 8 ranks itself: config 4 (12.5 M reads per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W    # same, ranks started by the launcher

A "step" is one pass of the hot path (pack windows -> adapter scan -> barcode scan -> finalize ->
count histogram, + one RCCL all-reduce of the count vector when N > 1) over one batch of
synthetic reads that is already resident in HBM.  Weak scaling: every rank owns its own shard
(distinct seed), value = N * reads_per_gpu * steps / (max over ranks of the timed region).

No PyTorch anywhere: one process per GPU over the C ABI of include/qcat_hip.h; the count vector is
all-reduced by RCCL inside the library (qcat_counts_allreduce), barrier and max-over-ranks go over
the same communicator, the RCCL unique id travels over a local TCP socket (qcat_amd/parallel.py).
--gpus N without a launcher environment starts the N ranks itself and fails if the node has fewer
than N devices.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel's ALGORITHMIC bytes
(150 B per scanned read end + 24 B result record) against the 8 TB/s HBM peak,
with the kernel's average duration measured by HIP events on the library's own stream;
`cpu_baseline` times the CPU oracle (a port, not the reference: parasail is absent) on a bounded
sample of the same reads on the host cores of this box (rank 0, at every N: the other ranks wait at
the final barrier).  `valu_issue` is the VALU view of the DP phases from COUNTERS, not from a model:
SQ_INSTS_VALU per launch (the newest profiles/r0N_pmc.json, rocprofv3 --pmc on this workload; replayed with a staleness guard) x 2 issue cycles /
(1024 SIMDs x the clock GRBM_GUI_ACTIVE measured x the phase time measured live in this run).

    python bench.py --workload api4000     # the reference driver's own call shape: detect_barcode_batch on
                                           # 4000-read batches, kit auto, results as Python dicts (cli.py:500-513)
"""
import argparse
import ctypes as C
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from qcat_amd import config as qconfig  # noqa: E402
from qcat_amd import native, parallel, scanner  # noqa: E402

WORKLOADS = {
    # name: (mode, kit, ends, tpl_5p, tpl_3p, algorithmic bytes per read, default reads per GPU)
    "config2": ("epi2me", "NBD103/NBD104", native.ENDS_5P, 1, 0, 174, 1000000),
    "config3": ("epi2me", "PBC096", native.ENDS_BOTH, 1, 0, 324, 10000000),
    "config4": ("epi2me", "PBC096", native.ENDS_BOTH, 1, 0, 324, 12500000),     # per GPU: 100 M over 8
    "dual": ("dual", None, native.ENDS_BOTH, 1, 0, 324, 1000000),
    "dual96": ("dual", None, native.ENDS_BOTH, 1, 0, 324, 1000000),             # custom kit, 96 x 96 pairs
    # --detect-middle (every called read's interior is scanned on both strands);
    # algorithmic bytes = both windows + the interior (~424 nt of a ~724-nt read) + the record
    "middle": ("epi2me", "NBD103/NBD104", native.ENDS_BOTH, 1, 0, 324 + 424, 1000000),
    # the reference driver's call shape (qcat/cli.py:500-513): kit auto (12 templates), detect_barcode_batch on batches
    # of 4000 reads, results as Python dicts; reads carry PBC096 adapters; host-driven, see api4000()
    "api4000": ("epi2me", None, native.ENDS_BOTH, 3, 2, 324, 200000),
    # the reference's library / test entry: detect_barcode on ONE read per call (qcat/test/test_barcode.py:84, :309-322;
    # cli.py:504-509 --no-batch), a named kit and kit auto (all twelve templates, no vote); host-driven, see api1()
    "api1": ("epi2me", None, native.ENDS_BOTH, 3, 2, 324, 2000),
}
HBM_PEAK_GBS = 8000.0


def usable_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the
    GPU boxes expose 256 hardware threads but run containers under a CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2: "<quota> <period>" or "max ..."
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (IOError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / per
        except (IOError, ValueError):
            quota = None
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: config3 at --gpus 1, config4 (12.5 M PBC096 reads per GPU) at --gpus N > 1")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step (default: the workload's own size)")
    ap.add_argument("--error-rate", type=float, default=0.08)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adapter-fraction", type=float, default=0.05, help="share of the synthetic reads that carry no adapter (their barcode regions are whole 150-base windows)")
    ap.add_argument("--no-host-inclusive", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0,
                    help="CPU time of the oracle sample (cpu_baseline + parity); short by default so that the GPU work is not a\n"
                         "sliver of the run the driver's gpu_busy samples see")
    ap.add_argument("--batch", type=int, default=4000, help="api4000: reads per detect_barcode_batch call (cli.py:500)")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if a.workload is None:
        a.workload = "config3" if a.gpus == 1 else "config4"
    if a.reads is None:
        a.reads = WORKLOADS[a.workload][6]
    if a.seed is None:
        a.seed = 20260928 + {"config2": 1, "config3": 2, "config4": 3, "dual": 4, "dual96": 4, "middle": 1, "api4000": 2, "api1": 2}[a.workload]
    return a


def make_scanner(workload, mode, kit_name, device):
    if workload == "dual96":
        import custom_kits                                   # tests/custom_kits.py: writes the kit folder
        folder = custom_kits.dual_96x96_folder()
        return scanner.factory(mode=mode, kit_folder=folder, device=device)
    return scanner.factory(mode=mode, kit=kit_name, device=device)


def main():
    a = parse()
    if "RANK" not in os.environ and a.gpus > 1:
        # no launcher: start one rank per GPU ourselves (the ranks re-enter main() below)
        n_dev = native.HipLibrary.get().lib.qcat_device_count()
        if n_dev < a.gpus:
            sys.exit("bench.py: --gpus %d but only %d HIP device(s) are visible" % (a.gpus, n_dev))
        sys.exit(parallel.launch(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank, local_rank, world = parallel.rank_env()
    if a.gpus != world:
        sys.exit("bench.py: --gpus %d does not match the launcher's WORLD_SIZE %d" % (a.gpus, world))
    hip = native.HipLibrary.get()
    lib = hip.lib
    n_dev = lib.qcat_device_count()
    if local_rank >= n_dev:
        sys.exit("bench.py: rank %d needs HIP device %d but only %d device(s) are visible" % (rank, local_rank, n_dev))

    mode, kit_name, ends, t5, t3, bytes_per_read, _ = WORKLOADS[a.workload]
    if a.workload == "api1":
        if world != 1:
            sys.exit("bench.py: --workload api1 is a single-process measurement")
        return api1(a, hip, lib)
    if a.workload == "api4000":
        if world != 1:
            sys.exit("bench.py: --workload api4000 is a single-process measurement")
        return api4000(a, hip, lib)
    det = make_scanner(a.workload, mode, kit_name, local_rank)
    cfg = qconfig.qcatConfig()
    desc = det.descriptor(qcat_config=cfg, ends=ends, scan_middle=(a.workload == "middle"))
    t_kit = time.perf_counter()
    kit = native.NativeKit(desc, jit=True)      # custom kits: generated kernels compiled (hipRTC) before the first scan
    kit_seconds = time.perf_counter() - t_kit
    ctx = native.NativeContext(local_rank)
    n_buckets = desc.n_count_buckets
    use_comm = world > 1 or "RANK" in os.environ           # launcher environment (also with one process)
    t_comm = time.perf_counter()

    def comm_stage(name):
        # start-up of N > 1 ranks dies in one of three places that look alike from outside: say where this rank is
        sys.stderr.write("[bench.py rank %d/%d +%.2f s] %s\n" % (rank, world, time.perf_counter() - t_comm, name))
        sys.stderr.flush()
    comm = parallel.init_comm(ctx, rank, world, trace=comm_stage if world > 1 else None) if use_comm else None

    sp = native.SynthParams(seed=a.seed + 1000003 * rank, n_reads=a.reads, insert_len=600, lead_min=5,
                            lead_max=40, error_rate=a.error_rate, no_adapter_fraction=a.no_adapter_fraction,
                            tpl_5p=t5, tpl_3p=t3)
    batch = C.c_void_p()
    hip.check(lib.qcat_batch_synthesize(ctx.handle, kit.handle, C.byref(sp), C.byref(batch)))
    nb = C.c_uint64()
    nr = C.c_uint32()
    hip.check(lib.qcat_batch_info(batch, C.byref(nr), C.byref(nb)))

    def step():
        hip.check(lib.qcat_scan_resident(ctx.handle, kit.handle, batch))
        if comm is not None:
            # the only cross-GPU exchange of the path: RCCL all-reduce, in place on the device-resident
            # int64 count vector, on the library's stream (ordered after this step's kernels and before
            # the next step's memset of the vector -- no host synchronisation)
            comm.allreduce_counts()

    def sync_all():
        hip.check(lib.qcat_ctx_synchronize(ctx.handle))
        if comm is not None:
            comm.barrier()                                   # drains every rank's stream, then meets

    for _ in range(a.warmup):
        step()
    hip.check(lib.qcat_ctx_set_timing(ctx.handle, 1))
    kernel_ms = {}
    names = (C.c_char_p * 16)()
    ms = (C.c_float * 16)()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()                                       # no host synchronisation between steps
    sync_all()
    elapsed = time.perf_counter() - t0
    # per-phase HIP-event times recorded inside the timed region (the library keeps one event set per
    # scan in a ring of 64 and averages over them)
    k = lib.qcat_ctx_last_timing(ctx.handle, names, ms, 16)
    for i in range(k):
        kernel_ms.setdefault(names[i].decode(), []).append(float(ms[i]))
    if comm is not None:
        elapsed = comm.allreduce([elapsed], native.REDUCE_MAX)[0]

    # ---- results of the last step: parity spot check + counts -------------------------------
    recs = np.zeros(a.reads, dtype=native.RESULT_DTYPE)
    hip.check(lib.qcat_ctx_fetch_results(ctx.handle, recs.ctypes.data, a.reads))
    total_counts = np.zeros(n_buckets, dtype=np.int64)          # after the all-reduce: the global histogram
    hip.check(lib.qcat_ctx_fetch_counts(ctx.handle, total_counts.ctypes.data, n_buckets))
    n_barcode_buckets = n_buckets - len(desc.kit_names) - 2      # [barcodes.., none][kits.., none][skipped]
    counts_total = int(total_counts[:n_barcode_buckets].sum())
    if counts_total != world * a.reads:
        sys.exit("bench.py: the count vector holds %d reads, expected %d" % (counts_total, world * a.reads))

    if rank == 0:
        value = world * a.reads * a.steps / elapsed
        avg = {k: float(np.mean(v)) for k, v in kernel_ms.items()}
        compute = {k: v for k, v in avg.items() if not k.startswith("rccl")}
        dom = max(compute, key=compute.get) if compute else None
        traffic, traffic_src = replayed_traffic(a, dom, avg)
        roof = None
        if dom:
            achieved = a.reads * bytes_per_read / (avg[dom] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": a.reads * bytes_per_read,
                    "avg_launch_ms": round(avg[dom], 4),
                    "kernels_avg_ms": {k: round(v, 4) for k, v in avg.items()},
                    "note": "integer DP (bit-sliced boolean planes for the barcode scan and, in big batches, the adapter scan; exact-integer fp16 / u16 lanes for the rest): "
                            "VALU-issue bound, not HBM bound (SURVEY.md 8d); see valu_issue"}
        out = {"metric": "reads/sec demultiplexed", "value": round(value, 1), "unit": "reads/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": "bit planes / int16 (the reference's int32 DP as differences of neighbouring cells in boolean planes: two per "
                        "difference for the barcode scan, four for the adapter scan of big batches; 16-bit lanes -- u16 / exact-integer "
                        "f16 -- for the rest; all proven exact per kit by range bounds at kit creation; kits outside the bounds run "
                        "the int32 kernel)",
               "data": "synthetic",
               "config": {"workload": "%s: %d synthetic reads/GPU, kit %s (%s), %s, error rate %.2f, "
                                      "~%d nt reads" % (a.workload, a.reads, kit_name or "DUAL", mode,
                                                        "5' end only" if ends == native.ENDS_5P else "5'+3' ends with trims",
                                                        a.error_rate, nb.value // max(1, a.reads)),
                          "reads_per_gpu": a.reads, "reads_total": world * a.reads, "kit": kit_name or "DUAL", "mode": mode,
                          "n_barcodes": [len(s) for s in (det.layouts[0].barcode_set_1, det.layouts[0].barcode_set_2) if s],
                          "parallelism": "reads sharded x%d" % world},
               "roofline": roof,
               "counts_total": counts_total,
               "count_allreduce": ("rccl (qcat_counts_allreduce): in place on the device count vector, %d int64 buckets, on the scan's stream"
                                   % n_buckets) if comm is not None else "single process"}
        info = kit.describe()
        out["kernels"] = {"static_templates": "%d/%d" % (info["n_static_templates"], info["n_templates"]),
                          "static_barcode_groups": "%d/%d" % (info["n_static_groups"], info["n_groups"]),
                          "bitsliced_barcode_groups": "%d/%d (%d with the target letters compiled in)"
                                                      % (info["bitslice_groups"] & 0xFFFF, info["n_groups"], info["bitslice_groups"] >> 16),
                          "kit_prepare_s": round(kit_seconds, 2)}
        if "rccl_counts_allreduce" in avg:
            out["count_allreduce_ms"] = round(avg["rccl_counts_allreduce"], 4)
        out["valu_issue"] = valu_issue(a, avg)
    if not a.no_host_inclusive:
        # PCIe-inclusive rate of the host-buffer entry point (never `value`): download the shard (at N > 1 its
        # first 2 M reads: N ranks x 9.5 GB of pageable host copies are not needed to see the rate), then time
        # qcat_scan_batch (upload + scan + 24 B/read download) three times, keep the fastest; at N > 1 all
        # ranks run it at the same time under the launcher's CPU split and the slowest rank counts
        hi = host_inclusive(a, hip, lib, ctx, kit, cfg, ends, batch, sp, nb.value, recs, comm, world, det, mode)
        if rank == 0:
            out["host_inclusive"] = hi
    if rank == 0:
        # ---- CPU baseline + parity on a bounded sample of rank 0's shard (every N) --------------
        if not a.no_cpu_baseline:
            cpu_legs(a, out, lib, kit, sp, desc, det, cfg, mode, recs)
        print(json.dumps(out))
        sys.stdout.flush()
    lib.qcat_batch_destroy(batch)
    if comm is not None:
        comm.barrier()
        comm.close()


def cgroup_cpu_stat():
    """(periods, throttled periods, throttled microseconds) of this container's CPU controller (cgroup v2 cpu.stat): whether
    the host threads of a leg were stalled by the CPU quota while it ran (None where the file is absent)."""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as fh:
            kv = dict(line.split() for line in fh if line.strip())
        return int(kv.get("nr_periods", 0)), int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except (IOError, ValueError):
        return None


def from_fastq(ctx, kit, det, mode, hb, ho, n, recs):
    """The same reads from a FASTQ FILE: qcat_fastq_open (mmap + record splitting on the host threads) and
    qcat_fastq_demux (scan straight from the mapping, TSV written beside the scan) on the first min(n, 1 M) reads,
    records compared with the resident scan's.  File-to-TSV rate of the native driver path (host-bound)."""
    import tempfile
    m = min(n, 1000000)
    tmp = tempfile.mkdtemp(prefix="qcat_bench_fq_")
    path = os.path.join(tmp, "reads.fastq")
    raw = hb[:int(ho[m])].tobytes()
    qual = b"I" * 65536
    # written in pieces of 16 k reads (~25 MB) -- the way a basecaller, `cat` or a copy writes a file: the page cache then holds it
    # in large folios.  A file written by a million 1.5 KB writes sits in single 4 KB pages, and every mapping operation on it
    # (populate, fault-around, zap) costs 3-10 x as much per byte: QCAT_BENCH_SMALL_WRITES=1 measures that case
    small_writes = os.environ.get("QCAT_BENCH_SMALL_WRITES") == "1"
    with open(path, "wb", buffering=0) as fh:
        for i0 in range(0, m, 16384):
            piece = []
            for i in range(i0, min(m, i0 + 16384)):
                s = raw[int(ho[i]):int(ho[i + 1])]
                piece.append(b"@r%d ch=%d\n%s\n+\n%s\n" % (i, 1 + i % 512, s, qual[:len(s)] if len(s) <= 65536 else b"I" * len(s)))
            if small_writes:
                for rec in piece:
                    fh.write(rec)
            else:
                fh.write(b"".join(piece))
    size = os.path.getsize(path)
    native.FastqFile(path).close()                   # (page cache warm, as after the file was just written)
    best = None
    with open(os.path.join(tmp, "calls.tsv"), "wb") as sink:
        for _ in range(2):                           # (the first call sizes the context's staging buffers)
            sink.seek(0)
            t1 = time.perf_counter()
            fq = native.FastqFile(path)
            got, _skipped, st = fq.demux(ctx, kit, det.layouts, mode == "dual", kit_auto=False, trim=True, min_read_length=0,
                                         tsv_fd=sink.fileno())
            dt = time.perf_counter() - t1
            fq.close()
            if best is None or dt < best[0]:
                best = (dt, st)
    same = got.tobytes() == recs[:m].tobytes()
    # the driver's own path since round 5: the file in segments through read | scan | write (qcat_fastq_demux_stream), host
    # memory independent of the file; its TSV must be the whole-file call's, its histogram the records'
    with open(os.path.join(tmp, "calls.tsv"), "rb") as fh:
        want_tsv = fh.read()
    stream_best = None
    stream_ok = True
    sink2 = open(os.path.join(tmp, "stream.tsv"), "w+b")
    throttle = None
    runs_ms = []
    for reader in (0,) * max(1, int(os.environ.get("QCAT_BENCH_STREAM_REPEATS", "3"))):     # (the library's default reader: mapped windows)
        sink2.seek(0)
        cs0 = cgroup_cpu_stat()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t1 = time.perf_counter()
        bc, ad, none, ad_none, st2 = native.FastqFile.demux_stream(path, ctx, kit, det.layouts, mode == "dual", kit_auto=False, trim=True,
                                                                    min_read_length=0, tsv_fd=sink2.fileno(), reader=reader)
        dt2 = time.perf_counter() - t1
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        cs1 = cgroup_cpu_stat()
        runs_ms.append(round(dt2 * 1e3, 2))
        sink2.seek(0)
        stream_ok = stream_ok and sink2.read(len(want_tsv) + 1) == want_tsv
        called = (got["barcode_idx"] >= 0) & (got["adapter_idx"] >= 0) & ((got["barcode2_idx"] >= 0) | (mode != "dual"))
        stream_ok = stream_ok and int(bc.sum()) == int(called.sum()) and none == m - int(called.sum()) and int(ad.sum()) + ad_none == m \
            and st2["n_reads"] == m and not st2["incomplete"]
        if stream_best is None or dt2 < stream_best[0]:
            stream_best = (dt2, st2, reader)
            throttle = {"cpu_s": round((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime), 3),
                        "of_which_system_s": round(ru1.ru_stime - ru0.ru_stime, 3),
                        "cgroup_periods": cs1[0] - cs0[0] if cs0 and cs1 else None,
                        "cgroup_throttled_periods": cs1[1] - cs0[1] if cs0 and cs1 else None,
                        "cgroup_throttled_ms": round((cs1[2] - cs0[2]) / 1e3, 2) if cs0 and cs1 else None}
    sink2.close()
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    if not same:
        sys.exit("bench.py: the scan of the FASTQ file and the resident scan disagree")
    if not stream_ok:
        sys.exit("bench.py: the streamed demux of the FASTQ file and the whole-file call disagree")
    dt, st = best
    dt2, st2, reader = stream_best
    return {"value": round(m / dt2, 1), "unit": "reads/s", "reads": m, "file_gb": round(size / 1e9, 3),
            "stream": {"reader": "mapped windows", "segments": st2["segments"], "runs_ms": runs_ms, "host": throttle,
                       "split_s": {k: round(st2[k], 4) for k in ("parse_s", "scan_s", "write_s", "total_s")}},
            "whole_file": {"value": round(m / dt, 1), "parse_gb_per_s": round(size / st["parse_s"] / 1e9, 2),
                           "split_s": {k: round(st[k], 4) for k in ("parse_s", "scan_s", "write_s", "total_s")}},
            "note": "value: qcat_fastq_demux_stream of a FASTQ file of the first reads, TSV out (one line per read) -- the file in "
                    "segments through read | scan | write, the three stages side by side (stream.split_s: busy time per stage); "
                    "whole_file: qcat_fastq_open + qcat_fastq_demux (index of the whole file, then the scan with the writers beside "
                    "it); TSV bytes identical between the two, records of the whole-file call identical to the resident scan's; host-bound"}


def from_fastq_sharded(det, cfg, hb, ho, n, comm, rank, world):
    """N > 1 (round 6): ONE FASTQ file over the ranks -- rank 0 writes the first min(n, 1 M) reads of its shard as a file, every
    rank takes its whole batches of 4000 reads (parallel.file_shards: qcat_fastq_batch_offsets), demultiplexes its byte range
    with a TSV of its own (qcat_fastq_demux_stream with range_begin / range_end), the histograms meet in one all-reduce and
    rank 0 strings the TSV shards together.  Strong scaling of a small file: what it shows is that the ranks can share a file,
    not a rate to compare with the resident one.  Never raises: a failure is reported in the line."""
    import shutil
    import tempfile
    try:
        m = min(n, 1000000)
        tmp = os.path.join(tempfile.gettempdir(), "qcat_bench_shared_" + parallel.job_nonce().hex())
        path = os.path.join(tmp, "reads.fastq")
        if rank == 0:
            os.makedirs(tmp, exist_ok=True)
            raw = hb[:int(ho[m])].tobytes()
            qual = b"I" * 65536
            with open(path, "wb", buffering=0) as fh:
                for i0 in range(0, m, 16384):
                    fh.write(b"".join(b"@r%d ch=%d\n%s\n+\n%s\n" % (i, 1 + i % 512, raw[int(ho[i]):int(ho[i + 1])],
                                                                       qual[:int(ho[i + 1] - ho[i])] if int(ho[i + 1] - ho[i]) <= 65536 else b"I" * int(ho[i + 1] - ho[i]))
                                      for i in range(i0, min(m, i0 + 16384))))
        comm.barrier()
        shards, n_reads, _ = parallel.file_shards(path, world)
        tsv = os.path.join(tmp, "calls.tsv")
        best = None
        for _ in range(2):                               # (the first call sizes the context's staging buffers)
            comm.barrier()
            t1 = time.perf_counter()
            res = parallel.demux_file_shard(det, path, rank, world, cfg, tsv_path=tsv, trim=True, comm=comm, shards=shards)
            dt = comm.allreduce([time.perf_counter() - t1], native.REDUCE_MAX)[0]
            best = dt if best is None else min(best, dt)
        out = None
        if rank == 0:
            parallel.merge_shards(tsv, world)
            with open(tsv, "rb") as fh:
                rows = sum(chunk.count(b"\n") for chunk in iter(lambda: fh.read(1 << 24), b""))
            ok = n_reads == m and res[4].get("n_reads_total", res[4]["n_reads"]) == m and rows == m
            out = {"value": round(m / best, 1), "unit": "reads/s", "reads": m, "ranks": world, "seconds": round(best, 4),
                   "shards_bytes": [e - s for s, e in shards], "rows_and_counts_complete": bool(ok),
                   "note": "one file of %d reads over %d ranks: whole batches of 4000 reads per rank, a TSV shard per rank, histograms "
                           "all-reduced, shards strung together by rank 0 (qcat_amd/parallel.py); max over the ranks" % (m, world)}
        comm.barrier()
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
        return out
    except Exception as e:                               # noqa: BLE001 -- a diagnostic leg must not take the bench line down
        return {"error": "%s: %s" % (type(e).__name__, e)} if rank == 0 else None


def host_inclusive(a, hip, lib, ctx, kit, cfg, ends, batch, sp, n_bases, recs, comm, world, det, mode):
    n = a.reads if world == 1 else min(a.reads, 2000000)
    small = None
    if n < a.reads:
        # the generator is stateless per read index: a batch of the first n reads of the same parameters IS the head
        # of this rank's shard
        sp2 = native.SynthParams.from_buffer_copy(sp)
        sp2.n_reads = n
        small = C.c_void_p()
        hip.check(lib.qcat_batch_synthesize(ctx.handle, kit.handle, C.byref(sp2), C.byref(small)))
        src, nb2, nr2 = small, C.c_uint64(), C.c_uint32()
        hip.check(lib.qcat_batch_info(src, C.byref(nr2), C.byref(nb2)))
        n_bases = nb2.value
    else:
        src = batch
    ho = np.zeros(n + 1, dtype=np.uint64)
    hb = np.zeros(n_bases, dtype=np.uint8)
    hip.check(lib.qcat_batch_download(ctx.handle, src, hb.ctypes.data, ho.ctypes.data))
    if small is not None:
        lib.qcat_batch_destroy(small)
    hip.check(lib.qcat_ctx_set_timing(ctx.handle, 0))
    hout = np.empty(n, dtype=native.RESULT_DTYPE)
    best = None
    for _ in range(3):                               # (the first call sizes the context's staging buffers)
        if comm is not None:
            comm.barrier()
        t1 = time.perf_counter()
        ctx.scan(kit, hb, ho, out=hout)
        dt = time.perf_counter() - t1
        if comm is not None:
            dt = comm.allreduce([dt], native.REDUCE_MAX)[0]
        best = dt if best is None else min(best, dt)
    if hout.tobytes() != recs[:n].tobytes():
        sys.exit("bench.py: the host-buffer scan and the resident scan disagree")
    keep = cfg.max_align_length * (1 if ends == native.ENDS_5P else 2)
    up = int(ho[n]) if a.workload == "middle" else int(np.minimum(np.diff(ho).astype(np.int64), keep).sum())
    # (the native FASTQ driver scans both ends, like qcat's own driver: workloads on a both-ends kit, one process)
    fq_leg = from_fastq(ctx, kit, det, mode, hb, ho, n, recs) if (world == 1 and ends == native.ENDS_BOTH and a.workload != "middle") else None
    if world > 1 and comm is not None and ends == native.ENDS_BOTH and a.workload != "middle":
        fq_leg = from_fastq_sharded(det, cfg, hb, ho, n, comm, parallel.rank_env()[0], world)
    return {"from_fastq": fq_leg, "value": round(world * n / best, 1), "unit": "reads/s", "reads_per_gpu": n, "host_threads_per_rank":
            int(os.environ.get("QCAT_HOST_THREADS", "0")) or min(usable_cores(), 16),
            "note": "qcat_scan_batch from pageable host memory (%.0f MB of reads per rank), records identical to the "
                    "resident scan's: chunks of 256 k to 1 M reads are compacted to their scanned windows on host "
                    "threads, uploaded and scanned as a three-stage pipeline; %.0f MB up, %.0f MB down per rank and call; "
                    "host-bound (DESIGN.md section 4)" % (int(ho[n]) / 1e6, up / 1e6, n * 24 / 1e6)}


STALE_TOLERANCE = 0.15


def newest_profile(suffix):
    """profiles/r0N_<suffix> of the latest round that has one"""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)):
        m = re.match(r"r(\d\d)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def stale_reason(rec, mark, live_ms):
    """Counters are REPLAYED from a committed rocprofv3 pass (the driver's run collects none): the pass says at which commit
    and at what duration of the mark it was taken, and a live duration more than 15 % off means the kernels are no longer
    the ones that were counted -- the figure is dropped then, with the reason, instead of being scaled into a wrong one."""
    ref = (rec.get("mark_ms") or {}).get(mark)
    if ref is None:
        return None                     # (files of rounds 1-3 carry no durations: nothing to compare with)
    if ref <= 0 or abs(live_ms - ref) > STALE_TOLERANCE * ref:
        return "stale: %s ran %.4f ms per launch when the counters were taken (commit %s), %.4f ms now" % (
            mark, ref, rec.get("commit", "?"), live_ms)
    return None


def replayed_traffic(a, dom, avg, path=None):
    wl = "config3" if a.workload == "config4" else a.workload
    path = path or newest_profile("traffic.json")
    if not path or not dom:
        return None, None
    try:
        with open(path) as fh:
            tj = json.load(fh).get(wl)
    except (IOError, ValueError):
        return None, None
    if not tj or dom not in tj.get("kernel", ""):
        return None, None
    scale = a.reads / float(tj["reads_per_launch"])
    why = stale_reason({"mark_ms": {dom: tj["mark_ms"] * scale} if tj.get("mark_ms") is not None else None, "commit": tj.get("commit")}, dom, avg[dom])
    rel = os.path.relpath(path, ROOT)
    if why:
        return None, "%s not replayed -- %s" % (rel, why)
    return int(tj["bytes"] * scale), "%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE at commit %s, scaled to this launch size)" % (rel, tj.get("commit", "?"))


def valu_issue(a, avg, path=None):
    """VALU-issue utilisation of the DP phases from hardware counters: instructions issued (SQ_INSTS_VALU, summed over
    the kernels of a timing mark, per launch, from the committed PMC pass of this workload, scaled to this launch's
    read count) x 2 cycles -- the fastest a wave64 VALU instruction issues on a CDNA SIMD -- / (1024 SIMDs x effective
    clock x the mark's duration measured live in this run).  No instruction-mix model and NO clamp: a fraction above 1
    cannot be a utilisation, so it is printed as null with the reason (a counter file that no longer belongs to the
    kernels), and so is a mark whose live duration is more than 15 % away from the one the counters were taken at."""
    wl = "config3" if a.workload == "config4" else a.workload
    path = path or newest_profile("pmc.json")
    try:
        with open(path) as fh:
            pj = json.load(fh).get(wl)
    except (IOError, ValueError, TypeError):
        pj = None
    if not pj:
        return None
    clock = float(pj["clock_ghz"]) * 1e9
    out = {"source": "%s (rocprofv3 --pmc SQ_INSTS_VALU / GRBM_GUI_ACTIVE at commit %s, tools/update_pmc.py), replayed: "
                     "durations are measured live" % (os.path.relpath(path, ROOT), pj.get("commit", "?")),
           "clock_ghz": pj["clock_ghz"], "simds": 1024, "issue_cycles_per_inst": 2, "marks": {}}
    scale = a.reads / float(pj["reads_per_launch"])
    for mark, m in pj["marks"].items():
        if mark not in avg or avg[mark] <= 0:
            continue
        insts = m["insts_valu"] * scale
        util = insts * 2.0 / (1024.0 * clock * avg[mark] * 1e-3)
        util24 = insts * 2.0 / (1024.0 * 2.4e9 * avg[mark] * 1e-3)
        rec = {"insts_valu_per_launch": int(insts), "ms": round(avg[mark], 4)}
        why = stale_reason({"mark_ms": {k: v * scale for k, v in pj["mark_ms"].items()} if pj.get("mark_ms") else None,
                            "commit": pj.get("commit")}, mark, avg[mark])
        if why is None and (util > 1.0 or util24 > 1.0):
            why = "impossible: %d instructions x 2 cycles do not fit %.4f ms on 1024 SIMDs (%.3f of the issue peak) -- the counter file does not belong to these kernels" % (int(insts), avg[mark], util)
        if why:
            rec.update({"issue_util": None, "issue_util_at_2p4ghz": None, "reason": why})
        else:
            rec.update({"issue_util": round(util, 4), "issue_util_at_2p4ghz": round(util24, 4)})
        out["marks"][mark] = rec
    return out


def cpu_legs(a, out, lib, kit, sp, desc, det, cfg, mode, recs):
    """cpu_baseline (the oracle on this box's host cores, bounded sample), parity of the HIP records
    against it, and the reference-defined DP cell rate (a plain rate, no ceiling)."""
    import oracle_lib
    ncpu = usable_cores()

    def sample_reads(n, index=None):
        buf = np.zeros(4096, dtype=np.uint8)
        chunks, offs = [], np.zeros(n + 1, dtype=np.uint64)
        for i in range(n):
            ln = lib.qcat_synth_read(kit.handle, C.byref(sp), i if index is None else int(index[i]), buf.ctypes.data, buf.size)
            chunks.append(buf[:ln].tobytes())
            offs[i + 1] = offs[i] + ln
        return np.frombuffer(b"".join(chunks), dtype=np.uint8), offs

    probe = min(2000, a.reads)
    packed = sample_reads(probe)
    t1 = time.perf_counter()
    o1, tr1 = oracle_lib.scan(desc, packed=packed, trace=True, threads=1)
    one_thread = probe / (time.perf_counter() - t1)
    # calibrate the all-core rate on a short run, then size the timed sample for ~cpu_seconds
    probe2 = int(min(a.reads, max(probe, one_thread * min(ncpu, 16) * 1.0)))
    packed = sample_reads(probe2)
    t1 = time.perf_counter()
    oracle_lib.scan(desc, packed=packed, threads=ncpu)
    rate_all = probe2 / (time.perf_counter() - t1)
    n_sample = int(min(a.reads, max(probe2, rate_all * a.cpu_seconds)))
    packed = sample_reads(n_sample)
    t1 = time.perf_counter()
    o = oracle_lib.scan(desc, packed=packed, threads=ncpu)
    all_cores = n_sample / (time.perf_counter() - t1)
    mism = int(np.count_nonzero(o != recs[:n_sample]))
    # ... and a RANDOM subset of the whole shard, another one in every run (the timed sample is always the same
    # first reads); the seed is reported
    rseed = int.from_bytes(os.urandom(4), "little")
    n_rand = int(min(a.reads, max(2000, min(n_sample, rate_all * 2.0))))
    ridx = np.sort(np.random.RandomState(rseed).choice(a.reads, size=n_rand, replace=False)) if n_rand < a.reads else np.arange(a.reads)
    o_rand = oracle_lib.scan(desc, packed=sample_reads(len(ridx), ridx), threads=ncpu)
    mism_rand = int(np.count_nonzero(o_rand != recs[ridx]))
    # reference-defined DP cells per read on the probe (second figure)
    lays = det.layouts
    cells_a = cells_b = 0
    tl = sum(l.get_adapter_length() for l in lays)
    for t in tr1:
        cells_a += int(t["window_len"]) * tl
        lay = lays[int(t["used_tpl"])]
        for s in range(2 if mode == "dual" else 1):
            bs = lay.get_barcode_set(s)
            tlen = (len(lay.get_upstream_context(cfg.barcode_context_length, s)) + len(bs[0].sequence)
                    + len(lay.get_downstream_context(cfg.barcode_context_length, s)))
            cells_b += len(bs) * int(t["region_len"][s]) * tlen
    cells_a /= float(probe)
    cells_b /= float(probe)
    region_frac = float(np.mean([int(t["region_path"]) for t in tr1])) if len(tr1) else 0.0
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = [l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")][0]
    except (IOError, IndexError):
        cpu_model = "unknown CPU"
    out["cpu_baseline"] = {"value": round(all_cores, 1), "unit": "reads/s", "cores": ncpu, "kind": "port",
                           "sample": "first %d reads of rank 0's shard, oracle/qcat_oracle.c with OpenMP over "
                                     "%d threads of %s (1 thread: %.0f reads/s on %d reads)" % (n_sample, ncpu, cpu_model, one_thread, probe)}
    out["parity"] = {"checked_reads": n_sample + len(ridx), "mismatches_vs_oracle": mism + mism_rand,
                     "first_reads": n_sample, "random_reads": int(len(ridx)), "random_seed": rseed, "mismatches_random": mism_rand}
    out["dp_cells"] = {"unit": "reference-defined DP cell updates (SURVEY.md 8d) -- a plain rate, no ceiling attached",
                       "adapter_cells_per_read": round(cells_a, 1), "barcode_cells_per_read": round(cells_b, 1),
                       "region_path_fraction": round(region_frac, 4),
                       "cell_updates_per_s": round((cells_a + cells_b) * out["value"], 1)}


def api4000(a, hip, lib):
    """The reference driver's own call shape (qcat/cli.py:500-513, scanner_base.py:714-733): detect_barcode_batch on
    batches of --batch reads (4000), kit auto (the 12 auto-detect templates: vote + detect_barcode of the voted kit),
    results as Python dicts.  Reads: synthetic PBC096 reads as Python strings, made before the timed region.  The split
    says where a batch's time goes: packing the strings, the native call, building the dicts."""
    det = scanner.factory(mode="epi2me", kit=None)
    cfg = qconfig.qcatConfig()
    gen = scanner.factory(mode="epi2me", kit="PBC096")
    gdesc = gen.descriptor(qcat_config=cfg, ends=native.ENDS_BOTH)
    gkit = native.NativeKit(gdesc)
    sp = native.SynthParams(seed=a.seed, n_reads=a.reads, insert_len=600, lead_min=5, lead_max=40,
                            error_rate=a.error_rate, no_adapter_fraction=a.no_adapter_fraction, tpl_5p=1, tpl_3p=0)
    buf = np.zeros(4096, dtype=np.uint8)
    reads = []
    for i in range(a.reads):
        ln = lib.qcat_synth_read(gkit.handle, C.byref(sp), i, buf.ctypes.data, buf.size)
        reads.append(buf[:ln].tobytes().decode("ascii"))
    batches = [reads[i:i + a.batch] for i in range(0, len(reads), a.batch)]
    quals = [[None] * len(b) for b in batches]
    for b, q in list(zip(batches, quals))[:max(1, a.warmup)]:
        det.detect_barcode_batch(b, q, cfg)
    split = {"pack_reads_s": 0.0, "native_call_s": 0.0, "dicts_s": 0.0}
    ctx = det._context()
    orig_scan_auto, orig_pack, orig_dicts = ctx.scan_auto, native.pack_reads, det._records_to_dicts
    orig_views, orig_scan_views = native.read_views, ctx.scan_auto_views

    def timed(key, fn):
        def wrap(*args, **kw):
            t = time.perf_counter()
            try:
                return fn(*args, **kw)
            finally:
                split[key] += time.perf_counter() - t
        return wrap
    ctx.scan_auto = timed("native_call_s", orig_scan_auto)
    ctx.scan_auto_views = timed("native_call_s", orig_scan_views)
    native.pack_reads = timed("pack_reads_s", orig_pack)
    native.read_views = timed("pack_reads_s", orig_views)     # (the str objects' buffers handed over as they are: csrc/pyglue.c)
    det._records_to_dicts = timed("dicts_s", orig_dicts)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        for b, q in zip(batches, quals):
            res = det.detect_barcode_batch(b, q, cfg)
    elapsed = time.perf_counter() - t0
    ctx.scan_auto, native.pack_reads, det._records_to_dicts = orig_scan_auto, orig_pack, orig_dicts
    native.read_views, ctx.scan_auto_views = orig_views, orig_scan_views
    # the sanity figure `called_fraction` comes from one more pass outside the timed region (counting 4000 dicts in Python is
    # 0.25 ms per call: the bench's own bookkeeping, not the call's)
    called = 0
    for b, q in zip(batches, quals):
        called += sum(1 for r in det.detect_barcode_batch(b, q, cfg) if r["barcode"] is not None)
    called *= a.steps
    n_calls = a.steps * len(batches)
    total = a.steps * len(reads)
    out = {"metric": "reads/sec demultiplexed", "value": round(total / elapsed, 1), "unit": "reads/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / n_calls * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "int16 / bit planes", "data": "synthetic",
           "config": {"workload": "api4000: detect_barcode_batch on %d-read batches (qcat/cli.py:500-513), kit auto "
                                  "(%d templates), results as Python dicts; %d synthetic PBC096 reads as Python strings; "
                                  "a step here is one batch call" % (a.batch, len(det.layouts), len(reads)),
                      "batch": a.batch, "calls": n_calls, "reads_total": total},
           "split_ms_per_call": {k[:-2] + "_ms": round(v / n_calls * 1e3, 4) for k, v in split.items()},
           "python_helper": "qcat_amd/_pyglue.so (read pointers + dicts in C)" if native._pyglue is not None else "none (pure-Python conversions)",
           "other_python_ms_per_call": round((elapsed - sum(split.values())) / n_calls * 1e3, 4),
           "called_fraction": round(called / float(total), 4),
           "roofline": None, "cpu_baseline": None,
           "note": "host-driven call shape: the GPU kernels of a 4000-read batch take a fraction of the call; "
                   "roofline / cpu_baseline belong to the resident workloads (default run)"}
    print(json.dumps(out))
    sys.stdout.flush()


def api1(a, hip, lib):
    """The reference's library entry (qcat/test/test_barcode.py:84, :309-322; the driver's --no-batch loop, cli.py:504-509):
    detect_barcode on ONE read per call, with a named kit (PBC096: two templates) and under kit auto (all twelve auto-detect
    templates, no vote).  Reads: synthetic PBC096 reads as Python strings.  The split says where a call's time goes."""
    cfg = qconfig.qcatConfig()
    gen = scanner.factory(mode="epi2me", kit="PBC096")
    gkit = native.NativeKit(gen.descriptor(qcat_config=cfg, ends=native.ENDS_BOTH))
    sp = native.SynthParams(seed=a.seed, n_reads=a.reads, insert_len=600, lead_min=5, lead_max=40,
                            error_rate=a.error_rate, no_adapter_fraction=a.no_adapter_fraction, tpl_5p=1, tpl_3p=0)
    buf = np.zeros(4096, dtype=np.uint8)
    reads = []
    for i in range(a.reads):
        ln = lib.qcat_synth_read(gkit.handle, C.byref(sp), i, buf.ctypes.data, buf.size)
        reads.append(buf[:ln].tobytes().decode("ascii"))
    legs = {}
    for name, det in (("named_kit", gen), ("kit_auto", scanner.factory(mode="epi2me", kit=None))):
        for r in reads[:max(8, a.warmup)]:
            det.detect_barcode(r, None, cfg)
        split = {"pack_reads_s": 0.0, "native_call_s": 0.0, "dicts_s": 0.0}
        ctx = det._context()
        orig_scan, orig_pack, orig_dicts = ctx.scan, native.pack_reads, det._records_to_dicts

        def timed(key, fn):
            def wrap(*args, **kw):
                t = time.perf_counter()
                try:
                    return fn(*args, **kw)
                finally:
                    split[key] += time.perf_counter() - t
            return wrap
        ctx.scan = timed("native_call_s", orig_scan)
        native.pack_reads = timed("pack_reads_s", orig_pack)
        det._records_to_dicts = timed("dicts_s", orig_dicts)
        t0 = time.perf_counter()
        called = 0
        for _ in range(a.steps):
            for r in reads:
                called += det.detect_barcode(r, None, cfg)["barcode"] is not None
        elapsed = time.perf_counter() - t0
        ctx.scan, native.pack_reads, det._records_to_dicts = orig_scan, orig_pack, orig_dicts
        n_calls = a.steps * len(reads)
        # the same reads through one batch call: the single-read calls must give the same results
        batch = det._run(reads, det.layouts, cfg)
        single = [det.detect_barcode(r, None, cfg) for r in reads[:200]]
        if any(s != b for s, b in zip(single, batch[:200])):
            sys.exit("bench.py: detect_barcode on single reads and on the batch disagree")
        legs[name] = {"ms_per_call": round(elapsed / n_calls * 1e3, 4), "calls_per_s": round(n_calls / elapsed, 1), "templates": len(det.layouts),
                      "split_ms_per_call": {k[:-2] + "_ms": round(v / n_calls * 1e3, 4) for k, v in split.items()},
                      "other_python_ms_per_call": round((elapsed - sum(split.values())) / n_calls * 1e3, 4),
                      "called_fraction": round(called / float(n_calls), 4)}
    out = {"metric": "reads/sec demultiplexed", "value": legs["named_kit"]["calls_per_s"], "unit": "reads/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": legs["named_kit"]["ms_per_call"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "int16 / bit planes", "data": "synthetic",
           "config": {"workload": "api1: detect_barcode on ONE read per call (qcat/test/test_barcode.py:84, cli.py:504-509), results as "
                                  "Python dicts; %d synthetic PBC096 reads as Python strings; value = the named-kit leg; a step here is "
                                  "one call" % len(reads), "calls": a.steps * len(reads)},
           "legs": legs, "roofline": None, "cpu_baseline": None,
           "note": "host-driven call shape: one read is two windows -- the kernels are latency, not throughput; roofline / cpu_baseline "
                   "belong to the resident workloads (default run)"}
    print(json.dumps(out))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
