"""The cross-GPU exchange of the path through the C ABI (include/qcat_hip.h: qcat_comm_*,
qcat_counts_allreduce -- RCCL inside the library, no PyTorch): world size 1 on any box, world
size 2 (two host threads, one context per device, and two processes started by the product's
launcher) when the box has two GPUs.  The all-reduced histogram must equal the oracle's counts of
the WHOLE batch (the vector of qcat/cli.py:366-383 / scanner_base.py:680-689)."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import oracle_lib
import synth
from qcat_amd import native, parallel, scanner

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scan_shard(ctx, kit, reads):
    """resident scan of one shard; the counts stay on the device"""
    hip = native.HipLibrary.get()
    bases, offsets = native.pack_reads(reads)
    batch = C.c_void_p()
    hip.check(hip.lib.qcat_batch_upload(ctx.handle, bases.ctypes.data, offsets.ctypes.data, len(reads), C.byref(batch)))
    hip.check(hip.lib.qcat_scan_resident(ctx.handle, kit.handle, batch))
    return batch


def _fetch_counts(ctx, n):
    hip = native.HipLibrary.get()
    out = np.zeros(n, dtype=np.int64)
    hip.check(hip.lib.qcat_ctx_fetch_counts(ctx.handle, out.ctypes.data, n))
    return out


@pytest.mark.parametrize("mode,kit_name", [("epi2me", "PBC096"), ("dual", None)])
def test_world1_counts_allreduce_is_the_identity(mode, kit_name):
    hip = native.HipLibrary.get()
    det = scanner.factory(mode=mode, kit=kit_name)
    desc = det.descriptor()
    kit = native.NativeKit(desc)
    ctx = native.NativeContext(0)
    reads = synth.synth_batch(500, 99, det.layouts, 1, 0, error_rate=0.08)
    _, want = oracle_lib.scan(desc, reads, counts=True, threads=8)
    comm = native.NativeComm(ctx, 1, 0, native.comm_unique_id())
    n_ranks, rank, dev = C.c_int(), C.c_int(), C.c_int()
    hip.check(hip.lib.qcat_comm_info(comm.handle, C.byref(n_ranks), C.byref(rank), C.byref(dev)))
    assert (n_ranks.value, rank.value, dev.value) == (1, 0, 0)
    batch = _scan_shard(ctx, kit, reads)
    comm.allreduce_counts()
    comm.allreduce_counts()                                   # SUM over one rank: still the same vector
    assert np.array_equal(_fetch_counts(ctx, desc.n_count_buckets), want)
    comm.barrier()
    assert comm.allreduce([1.5, -2.0], native.REDUCE_SUM) == [1.5, -2.0]
    assert comm.allreduce([3.25], native.REDUCE_MAX) == [3.25]
    with pytest.raises(RuntimeError):
        comm.allreduce([0.0] * 65)
    hip.lib.qcat_batch_destroy(batch)
    comm.close()


def test_allreduce_needs_a_scan_and_matching_devices():
    ctx = native.NativeContext(0)
    comm = native.NativeComm(ctx, 1, 0, native.comm_unique_id())
    with pytest.raises(RuntimeError, match="no scan"):
        comm.allreduce_counts()
    with pytest.raises(RuntimeError, match="rank"):
        native.NativeComm(ctx, 2, 2, native.comm_unique_id())
    comm.close()


def _two_gpus():
    return native.HipLibrary.get().lib.qcat_device_count() >= 2


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs")
def test_world2_threads_one_context_per_device():
    det = scanner.factory(kit="PBC096")
    desc = det.descriptor()
    kit = native.NativeKit(desc)                              # one kit shared by both devices
    n = 2001
    reads = synth.synth_batch(n, 7, det.layouts, 1, 0, error_rate=0.08)
    _, want = oracle_lib.scan(desc, reads, counts=True, threads=8)
    uid = native.comm_unique_id()
    got, errs = [None, None], []

    def rank_main(r):
        try:
            ctx = native.NativeContext(r)
            comm = native.NativeComm(ctx, 2, r, uid)
            b, e = parallel.shard_range(n, r, 2)
            batch = _scan_shard(ctx, kit, reads[b:e])
            comm.allreduce_counts()
            got[r] = _fetch_counts(ctx, desc.n_count_buckets)
            assert comm.allreduce([float(r + 1)], native.REDUCE_MAX) == [2.0]
            comm.barrier()
            native.HipLibrary.get().lib.qcat_batch_destroy(batch)
            comm.close()
        except Exception as ex:                                # noqa: BLE001
            errs.append(ex)

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs")
def test_bench_starts_two_ranks_itself():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--reads", "200000"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["counts_total"] == 400000
    assert line["count_allreduce"].startswith("rccl")
    assert "config4" in line["config"]["workload"]


def test_bench_under_a_launcher_environment_with_one_rank():
    """RANK/WORLD_SIZE = 0/1 (what torch.distributed.run sets for one process): the communicator
    path runs, RCCL included, on a one-GPU box."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(parallel.free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--reads", "100000", "--cpu-seconds", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["counts_total"] == 100000
    assert line["count_allreduce"].startswith("rccl") and "count_allreduce_ms" in line
    assert "config3" in line["config"]["workload"]
    # a line printed under a launcher carries everything a single-process line does (the N > 1 lines of the
    # scaling run are printed by this code path)
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] <= 1
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    assert line["parity"]["mismatches_vs_oracle"] == 0 and line["host_inclusive"]["value"] > 0
    assert_no_fraction_above_one(line)


def test_rehearsal_of_eight_ranks_on_one_device():
    """`bench.py --gpus 8` end to end on a ONE-GPU box (tests/rehearse_gpus8.py): the product's launcher starts 8 ranks
    (NUMA / CPU-quota plan, QCAT_HOST_THREADS), every rank runs the unmodified bench.main() on device 0 with the
    communicator stubbed over TCP (RCCL refuses several ranks per device): distinct seeds per rank, 1 M reads per shard,
    the count vector summed over the ranks, barrier + max-over-ranks timing, host_inclusive at usable CPUs / 8 host
    threads per rank, cpu_baseline + parity on rank 0, ONE JSON line with n_gpus = 8.  The value is that of eight
    processes sharing one GPU -- the rehearsal proves the code path, not a scaling figure."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rehearse_gpus8.py"), "--launch", "--gpus", "8",
                        "--reads", "1000000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["counts_total"] == 8000000 and line["scaling"] == "weak"
    assert "config4" in line["config"]["workload"] and line["config"]["reads_total"] == 8000000
    assert line["config"]["parallelism"] == "reads sharded x8"
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] <= 1
    assert line["cpu_baseline"]["value"] > 0 and line["parity"]["mismatches_vs_oracle"] == 0
    hi = line["host_inclusive"]
    assert hi["value"] > 0 and hi["reads_per_gpu"] == 1000000
    # round 6: ONE file over the eight ranks (whole batches per rank, a TSV shard each, histograms all-reduced, shards merged)
    fq = hi["from_fastq"]
    assert "error" not in fq, fq
    assert fq["ranks"] == 8 and fq["reads"] == 1000000 and fq["rows_and_counts_complete"] is True and len(fq["shards_bytes"]) == 8
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    assert hi["host_threads_per_rank"] == max(1, bench.usable_cores() // 8)
    assert_no_fraction_above_one(line)
    print("rehearsal: %.1f s, value %.1f M reads/s (8 ranks on ONE GPU), host_inclusive %.1f M reads/s at %d host threads per rank"
          % (time.time() - t0, line["value"] / 1e6, hi["value"] / 1e6, hi["host_threads_per_rank"]))


def assert_no_fraction_above_one(obj, path="line"):
    """every `frac` / `*_util*` field of the JSON line is a fraction of a hardware limit: never above 1"""
    if isinstance(obj, dict):
        for k, v in obj.items():
            if isinstance(v, (int, float)) and (k == "frac" or "util" in k or k.startswith("frac_")):
                assert 0 <= v <= 1.0, "%s.%s = %r" % (path, k, v)
            assert_no_fraction_above_one(v, path + "." + k)
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            assert_no_fraction_above_one(v, "%s[%d]" % (path, i))
