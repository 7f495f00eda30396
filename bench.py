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
sample of the same reads on the host cores of this box.
"""
import argparse
import ctypes as C
import json
import os
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
    ap.add_argument("--no-host-inclusive", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if a.workload is None:
        a.workload = "config3" if a.gpus == 1 else "config4"
    if a.reads is None:
        a.reads = WORKLOADS[a.workload][6]
    if a.seed is None:
        a.seed = 20260928 + {"config2": 1, "config3": 2, "config4": 3, "dual": 4, "dual96": 4, "middle": 1}[a.workload]
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
    det = make_scanner(a.workload, mode, kit_name, local_rank)
    cfg = qconfig.qcatConfig()
    desc = det.descriptor(qcat_config=cfg, ends=ends, scan_middle=(a.workload == "middle"))
    t_kit = time.perf_counter()
    kit = native.NativeKit(desc, jit=True)      # custom kits: generated kernels compiled (hipRTC) before the first scan
    kit_seconds = time.perf_counter() - t_kit
    ctx = native.NativeContext(local_rank)
    n_buckets = desc.n_count_buckets
    use_comm = world > 1 or "RANK" in os.environ           # launcher environment (also with one process)
    comm = parallel.init_comm(ctx, rank, world) if use_comm else None

    sp = native.SynthParams(seed=a.seed + 1000003 * rank, n_reads=a.reads, insert_len=600, lead_min=5,
                            lead_max=40, error_rate=a.error_rate, no_adapter_fraction=0.05,
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
        traffic = traffic_src = None
        for tname in ("r02_traffic.json", "r01_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", tname)) as fh:
                    tj = json.load(fh).get("config3" if a.workload == "config4" else a.workload)
                if tj and dom and dom in tj["kernel"]:
                    traffic = int(tj["bytes"] * (a.reads / float(tj["reads_per_launch"])))
                    traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, scaled to this launch size)" % tname
                    break
            except (IOError, ValueError, KeyError):
                continue
        roof = None
        if dom:
            achieved = a.reads * bytes_per_read / (avg[dom] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": a.reads * bytes_per_read,
                    "avg_launch_ms": round(avg[dom], 4),
                    "kernels_avg_ms": {k: round(v, 4) for k, v in avg.items()},
                    "note": "integer DP (bit-sliced boolean planes for the barcode scan, exact-integer fp16 / u16 lanes for the rest): "
                            "VALU-issue bound, not HBM bound (SURVEY.md 8d); see valu"}
        out = {"metric": "reads/sec demultiplexed", "value": round(value, 1), "unit": "reads/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": "int16 / bit planes (the reference's int32 DP as differences of neighbouring cells in two boolean planes "
                        "for the barcode scan, in 16-bit lanes -- u16 / exact-integer f16 -- for the adapter scan; both proven exact "
                        "per kit by range bounds at kit creation; kits outside the bounds run the int32 kernel)",
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
        if not a.no_host_inclusive and world == 1:
            # PCIe-inclusive rate of the host-buffer entry point (never `value`): download the shard,
            # then time qcat_scan_batch (upload + scan + 24 B/read download) twice, keep the faster.
            hb = np.zeros(nb.value, dtype=np.uint8)
            ho = np.zeros(a.reads + 1, dtype=np.uint64)
            hip.check(lib.qcat_batch_download(ctx.handle, batch, hb.ctypes.data, ho.ctypes.data))
            hip.check(lib.qcat_ctx_set_timing(ctx.handle, 0))
            best = None
            hout = np.empty(a.reads, dtype=native.RESULT_DTYPE)
            for _ in range(3):                           # (the first call sizes the context's staging buffers)
                t1 = time.perf_counter()
                ctx.scan(kit, hb, ho, out=hout)
                dt = time.perf_counter() - t1
                best = dt if best is None else min(best, dt)
            if hout.tobytes() != recs.tobytes():
                sys.exit("bench.py: the host-buffer scan and the resident scan disagree")
            keep = cfg.max_align_length * (1 if ends == native.ENDS_5P else 2)
            up = nb.value if a.workload == "middle" else int(np.minimum(np.diff(ho).astype(np.int64), keep).sum())
            out["host_inclusive"] = {"value": round(a.reads / best, 1), "unit": "reads/s",
                                     "note": "qcat_scan_batch from pageable host memory (%.0f MB of reads), records identical to the "
                                             "resident scan's: chunks of 256 k to 1 M reads are compacted to their scanned windows on host "
                                             "threads, uploaded and scanned as a three-stage pipeline; %.0f MB up, %.0f MB down per step"
                                             % (nb.value / 1e6, up / 1e6, a.reads * 24 / 1e6)}
            del hb, ho, hout

        # ---- CPU baseline + parity on a bounded sample of rank 0's shard ---------------------
        if not a.no_cpu_baseline and world == 1:      # the CPU legs run on rank 0 at N = 1 only
            cpu_legs(a, out, lib, kit, sp, desc, det, cfg, mode, recs, avg, elapsed)
        if world > 1:
            out.setdefault("cpu_baseline", None)        # measured at N = 1 (the driver's first run)
        print(json.dumps(out))
        sys.stdout.flush()
    lib.qcat_batch_destroy(batch)
    if comm is not None:
        comm.barrier()
        comm.close()


def cpu_legs(a, out, lib, kit, sp, desc, det, cfg, mode, recs, avg, elapsed):
    """cpu_baseline (the oracle on this box's host cores, bounded sample), parity of the HIP records
    against it, and the VALU-issue view of the two DP kernels."""
    import oracle_lib
    ncpu = usable_cores()

    def sample_reads(n):
        buf = np.zeros(4096, dtype=np.uint8)
        chunks, offs = [], np.zeros(n + 1, dtype=np.uint64)
        for i in range(n):
            ln = lib.qcat_synth_read(kit.handle, C.byref(sp), i, buf.ctypes.data, buf.size)
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
    out["parity"] = {"checked_reads": n_sample, "mismatches_vs_oracle": mism}
    # VALU-issue ceilings: one wave retires 128 cells per column; cycles per column from the issue
    # rates tools/valu_rate.hip measures on this chip (profiles/r01_valu_issue_rates.txt)
    simd_hz = 256 * 4 * 2.4e9
    bs_static = (kit.describe()["bitslice_groups"] >> 16) > 0
    cyc = {"k_adapter_packed": 4.17 + 2.73 + 4.15 + 4.15,    # v_perm_b32 + v_add_u32 + 2 x v_pk_max_u16 (u16 lanes)
           "k_barcode_packed": 4.17 + 4.18 + 4.20,           # v_perm_b32 + v_pk_add_f16 + v_pk_maximum3_f16
           "k_adapter_static": 4.18 + 4.20,                  # v_pk_add_f16 + v_pk_maximum3_f16 (static letters)
           "k_barcode_static": 4.18 + 4.20,
           # bit-sliced: 2048 cells per wave-column = 16 x 128; seven v_bitop3_b32 with VGPR sources at 2.04 cycles
           # (sustained, tools/valu_bank.hip), + two with an SGPR source at 4.15 when the letters come from memory
           "k_barcode_bitslice": (7 * 2.04 + (0 if bs_static else 2 * 4.15)) / 16.0}
    valu = {"unit": "DP cell updates/s", "dp_cells_per_read": round(cells_a + cells_b, 1),
            "region_path_fraction": round(region_frac, 4)}
    ideal_s = 0.0
    for name, cells, kerns in (("adapter", cells_a, ("k_adapter_static", "k_adapter_packed")),
                               ("barcode", cells_b, ("k_barcode_bitslice", "k_barcode_static", "k_barcode_packed"))):
        ran = [kn for kn in kerns if kn in avg]
        if not ran:
            continue
        ms = sum(avg[kn] for kn in ran)
        # the bit-sliced kernels take all but the odd lengths and the last jobs of a class: the phase is priced at
        # their rate when they ran; otherwise (mixed kits) at the slower instruction mix
        ceil = simd_hz * 128 / (cyc["k_barcode_bitslice"] if "k_barcode_bitslice" in ran else max(cyc[kn] for kn in ran))
        per_launch = cells * a.reads
        ideal_s += per_launch / ceil
        valu[name] = {"kernels": ran, "cells_per_read": round(cells, 1), "kernel_ms": round(ms, 4), "ceiling": round(ceil, 1),
                      "achieved": round(per_launch / (ms * 1e-3), 1), "frac": round(per_launch / (ms * 1e-3) / ceil, 4)}
    valu["frac_of_valu_peak"] = round(ideal_s / (elapsed / a.steps), 4)
    valu["note"] = ("ceiling = 1024 SIMDs x 2.4 GHz x 128 cells per wave-column / VALU issue cycles per column "
                    "(issue rates measured by tools/valu_rate.hip and tools/valu_bank.hip, profiles/r01_valu_issue_rates.txt, "
                    "profiles/r02_valu_operand_rates.txt).  Bit-sliced barcode kernels: seven v_bitop3_b32 per 2048 cells at 2.04 "
                    "cycles (+ two SGPR-source ones at 4.15 when the letters come from memory) = 0.89 (1.41) cycles per 128 cells; "
                    "static-letter fp16 kernels v_pk_add_f16 + v_pk_maximum3_f16 = 8.38 cycles; table kernels 12.55 (fp16 lanes) / "
                    "15.2 (u16 lanes); frac_of_valu_peak = ideal DP time of both phases / whole step time.  Cells are the "
                    "reference-defined ones (SURVEY.md 8d); the bit-sliced kernels compute the longer context's columns once per "
                    "2048 alignments (11 of PBC096's 42), the fp16 chains the columns their two or four targets share, so "
                    "`barcode.frac` counts more cells than the kernels evaluate")
    out["valu"] = valu


if __name__ == "__main__":
    main()
