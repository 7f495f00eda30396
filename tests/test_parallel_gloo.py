"""N > 1 path on CPU (no GPU here): world_size-2 processes shard a batch by read range, each rank
produces the count vector of its shard (with the CPU oracle -- the GPU kernels and the RCCL entry
point qcat_counts_allreduce are covered by the -m gpu tests in test_comm_gpu.py), the vectors are
all-reduced over gloo and must equal the counts of the whole batch.  The product's own multi-rank
plumbing -- shard arithmetic, launcher environment, the TCP rendezvous that carries the RCCL unique
id, the rank launcher -- is exercised for real; only the collective itself is gloo instead of RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib
import synth
from qcat_amd import native, parallel, scanner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(10, 2, 2)


def test_rank_env():
    assert parallel.rank_env({}) == (0, 0, 1)
    assert parallel.rank_env({"RANK": "3", "WORLD_SIZE": "8"}) == (3, 3, 8)
    assert parallel.rank_env({"RANK": "5", "LOCAL_RANK": "1", "WORLD_SIZE": "8"}) == (5, 1, 8)


# torch is imported inside the tests only: it bundles its own copy of the ROCm runtime, which must not
# be loaded into a process that runs the product's GPU tests (pytest imports every module it collects)


def _worker(rank, world, port, n_reads, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    # the product's rendezvous: rank 0 makes an id (RCCL's on a GPU box, random bytes here), all get it
    uid = parallel.exchange_id(rank, world, lambda: os.urandom(native.COMM_ID_BYTES))
    assert len(uid) == native.COMM_ID_BYTES
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = [None] * world
    dist.all_gather_object(ids, uid)
    assert all(i == uid for i in ids)                        # every rank holds rank 0's id
    det = scanner.factory(kit="PBC096")
    desc = det.descriptor()
    b, e = parallel.shard_range(n_reads, rank, world)
    reads = synth.synth_batch(e - b, 4711, det.layouts, 1, 0, first=b, error_rate=0.08)
    _, cnt = oracle_lib.scan(desc, reads, counts=True)
    t = torch.from_numpy(cnt.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                 # stands in for qcat_counts_allreduce (RCCL)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_count_allreduce_world2(tmp_path):
    import torch.multiprocessing as mp
    n = 240
    mp.spawn(_worker, args=(2, parallel.free_port(), n, str(tmp_path)), nprocs=2, join=True)
    det = scanner.factory(kit="PBC096")
    reads = synth.synth_batch(n, 4711, det.layouts, 1, 0, error_rate=0.08)
    _, want = oracle_lib.scan(det.descriptor(), reads, counts=True)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, want)
    assert want[:96].sum() + want[96] == n          # every read lands in exactly one barcode bucket


_RANK_SCRIPT = """
import os, sys
sys.path.insert(0, {root!r})
from qcat_amd import native, parallel
rank, local_rank, world = parallel.rank_env()
uid = parallel.exchange_id(rank, world, lambda: bytes(range(128)))
open(os.path.join({out!r}, "r%d" % rank), "wb").write(uid + bytes([rank, local_rank, world]))
sys.exit(int(os.environ.get("FAIL_RANK", "-1")) == rank)
"""


def test_launcher_starts_one_rank_per_gpu_and_propagates_failure(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT, out=str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "QCAT_RDZV_PORT")}
    assert parallel.launch(3, [str(script)], environ=env) == 0
    for r in range(3):
        data = (tmp_path / ("r%d" % r)).read_bytes()
        assert data[:128] == bytes(range(128)) and list(data[128:]) == [r, r, 3]
    env["FAIL_RANK"] = "1"
    assert parallel.launch(2, [str(script)], environ=env) != 0


def test_exchange_skips_a_foreign_server_on_the_first_candidate_port(tmp_path):
    """MASTER_PORT-derived candidates: a port held by somebody else is skipped by rank 0 and told
    apart by the other ranks (no answer / wrong magic)."""
    import socket
    base = parallel.free_port()
    squat = socket.socket()
    squat.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        squat.bind(("127.0.0.1", base + 1))
        squat.listen(4)
    except OSError:
        pytest.skip("neighbouring port not available")
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT, out=str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("QCAT_RDZV_PORT",)}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(base), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e))
    try:
        for p in procs:
            assert p.wait(timeout=120) == 0
    finally:
        squat.close()
    for r in range(2):
        assert (tmp_path / ("r%d" % r)).read_bytes()[:128] == bytes(range(128))


def test_bench_refuses_more_gpus_than_devices():
    """bench.py --gpus N must fail, not silently measure fewer devices (no GPU in this container)."""
    if native.HipLibrary.get().lib.qcat_device_count() >= 2:
        pytest.skip("multi-GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0
    assert b"HIP device" in p.stderr
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="4")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"WORLD_SIZE" in p.stderr


def test_two_jobs_of_one_world_size_do_not_cross_connect(tmp_path):
    """Per-job nonce in the rendezvous: a rank of job B that probes job A's server (same world size, neighbouring
    port) gets no id and is not counted by A; B's own server answers it."""
    import socket
    import struct
    import threading
    base = parallel.free_port()
    env_a = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(base), "QCAT_RDZV_NONCE": "job-a"}
    env_b = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(base), "QCAT_RDZV_NONCE": "job-b"}
    assert parallel.job_nonce(env_a) != parallel.job_nonce(env_b) and len(parallel.job_nonce(env_a)) == 8
    got = {}

    def run(tag, env, rank, payload):
        try:
            got[(tag, rank)] = parallel.exchange_id(rank, 2, lambda: payload, environ=env, timeout=60.0)
        except Exception as exc:                      # noqa: BLE001 -- reported through the assertion below
            got[(tag, rank)] = exc
    ts = [threading.Thread(target=run, args=("a", env_a, 0, b"A" * 128))]
    ts[0].start()
    import time
    time.sleep(0.3)                                   # A's server holds the first candidate port
    ts += [threading.Thread(target=run, args=("b", env_b, 0, b"B" * 128)),
           threading.Thread(target=run, args=("b", env_b, 1, None))]
    for t in ts[1:]:
        t.start()
    time.sleep(0.5)
    ts.append(threading.Thread(target=run, args=("a", env_a, 1, None)))
    ts[-1].start()
    for t in ts:
        t.join(timeout=90)
    assert got[("a", 0)] == got[("a", 1)] == b"A" * 128
    assert got[("b", 0)] == got[("b", 1)] == b"B" * 128
    # a request without the nonce (a rank of an older build, a port scanner) is not served either
    assert struct.calcsize("<ii") == 8 and socket is not None


def test_a_rank_whose_read_of_the_id_failed_is_served_again(tmp_path):
    """A client that drops the connection before it has read (and acknowledged) the id asks again: rank 0 serves it a
    second time and counts the rank once -- it used to skip a rank it had already answered, and the rank hung until the
    deadline."""
    import socket
    import struct
    import threading
    import time
    base = parallel.free_port()
    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(base), "QCAT_RDZV_NONCE": "job-retry"}
    got = {}

    def run(rank, payload):
        try:
            got[rank] = parallel.exchange_id(rank, 3, lambda: payload, environ=env, timeout=60.0)
        except Exception as exc:                      # noqa: BLE001
            got[rank] = exc
    srv = threading.Thread(target=run, args=(0, b"Z" * 128))
    srv.start()
    time.sleep(0.3)
    # rank 1's first attempt: sends a valid request, then closes without reading the reply
    nonce = parallel.job_nonce(env)
    with socket.create_connection(("127.0.0.1", parallel._candidate_ports(env)[0]), timeout=5.0) as conn:
        conn.sendall(parallel._MAGIC_REQ + nonce + struct.pack("<ii", 3, 1))
    t0 = time.time()
    ts = [threading.Thread(target=run, args=(r, None)) for r in (1, 2)]
    for t in ts:
        t.start()
    for t in ts + [srv]:
        t.join(timeout=90)
    assert got[0] == got[1] == got[2] == b"Z" * 128
    assert time.time() - t0 < 30.0


def test_rank_cpu_plan_splits_nodes_and_quota():
    """launch(): rank r sits on the CPUs of its GPU's NUMA node, ranks sharing a node split it, and the host threads
    per rank are the container's usable CPUs / ranks (16-core quota, 8 ranks -> 2 threads each, no oversubscription)."""
    node_cpus = {0: range(0, 64), 1: range(64, 128)}
    plan = parallel.rank_cpu_plan(8, numa_nodes=[0, 0, 0, 0, 1, 1, 1, 1], affinity=range(128), quota=16.0, node_cpus=node_cpus)
    assert [t for _c, t in plan] == [2] * 8
    seen = set()
    for r, (cpus, _t) in enumerate(plan):
        assert len(cpus) == 16 and set(cpus) <= set(node_cpus[r // 4]) and not (set(cpus) & seen)
        seen |= set(cpus)
    # unknown topology: the whole affinity mask, threads still split
    plan = parallel.rank_cpu_plan(4, numa_nodes=None, affinity=range(8), quota=None)
    assert [len(c) for c, _t in plan] == [8] * 4 and [t for _c, t in plan] == [2] * 4
    # more ranks than CPUs: one thread each, never zero
    assert [t for _c, t in parallel.rank_cpu_plan(8, affinity=range(2), quota=1.0)] == [1] * 8


def test_bench_gpus8_on_a_small_box_fails_fast():
    """bench.py --gpus 8 where fewer than 8 devices are visible: the device-count message, quickly, no rank started."""
    import time
    if native.HipLibrary.get().lib.qcat_device_count() >= 8:
        pytest.skip("8-GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"--gpus 8 but only" in p.stderr and p.stdout.strip() == b""
    assert time.time() - t0 < 20.0            # (seconds of interpreter + library start, no scan: 2 s on a warm box)


def test_comm_start_up_failure_names_rank_and_stage(monkeypatch):
    """bench.py --gpus N: a rank that dies while the communicator is built says which of rendezvous /
    ncclCommInitRank / first all-reduce it was in (parallel.init_comm), and traces the stages it entered."""
    from qcat_amd import native, parallel

    seen = []
    monkeypatch.setattr(parallel, "exchange_id", lambda *a, **k: (_ for _ in ()).throw(OSError("no route")))
    # (the exception keeps its type -- a caller that catches the rendezvous' OSError / TimeoutError still does -- and gains rank and stage)
    with pytest.raises(OSError, match=r"rank 3 of 8 failed in stage 'rendezvous': no route"):
        parallel.init_comm(None, 3, 8, environ={}, trace=seen.append)
    assert seen == ["rendezvous"]

    monkeypatch.setattr(parallel, "exchange_id", lambda *a, **k: b"\0" * 128)

    class Dead:
        def __init__(self, *a):
            raise RuntimeError("ncclCommInitRank: unhandled system error")
    monkeypatch.setattr(native, "NativeComm", Dead)
    seen.clear()
    with pytest.raises(RuntimeError, match=r"rank 1 of 2 failed in stage 'ncclCommInitRank'"):
        parallel.init_comm(None, 1, 2, environ={}, trace=seen.append)
    assert seen == ["rendezvous", "ncclCommInitRank"]

    closed = []

    class Short:
        def __init__(self, *a):
            pass

        def allreduce(self, values, op=0):
            return [1.0]

        def close(self):
            closed.append(True)
    monkeypatch.setattr(native, "NativeComm", Short)
    seen.clear()
    with pytest.raises(RuntimeError, match=r"failed in stage 'first all-reduce': all-reduce of one per rank gave 1.0 for 2 ranks"):
        parallel.init_comm(None, 0, 2, environ={}, trace=seen.append)
    assert seen == list(parallel.COMM_STAGES[:3])
    assert closed == [True]                   # a communicator that failed its proving all-reduce is closed, not leaked
