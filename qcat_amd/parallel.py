"""Multi-GPU plumbing of the hot path: reads are independent for a fixed kit, so a
batch is sharded by contiguous read ranges, one rank per GPU, with NO data-path collective; the
only exchange is one all-reduce (SUM, int64) of the per-barcode / per-kit count vector, done by
RCCL inside the native library (``qcat_counts_allreduce``, include/qcat_hip.h) in place on the
device-resident vector.

This module is the host side of that: the shard arithmetic, the rendezvous that carries the
128-byte RCCL unique id from rank 0 to the other ranks (a TCP socket on the node -- no PyTorch, no
MPI), and a launcher that starts one process per GPU.  Nothing here touches the kernels.
"""
import hashlib
import os
import socket
import struct
import subprocess
import sys
import time
import uuid

from . import native

_MAGIC_REQ = b"QCATRDZV2"
_MAGIC_REP = b"QCATID002"
_ACK = b"K"                                   # a rank acknowledges the id it has read: rank 0 counts it then
_N_CANDIDATE_PORTS = 16
_NONCE_BYTES = 8


def job_nonce(environ=None):
    """Eight bytes that tell this job's rendezvous from another job's on the same node: derived from
    QCAT_RDZV_NONCE (set by :func:`launch`) or, under a foreign launcher, from its run id and MASTER_PORT
    (every rank of one job sees the same values, two jobs of one node cannot share a MASTER_PORT)."""
    env = os.environ if environ is None else environ
    src = env.get("QCAT_RDZV_NONCE") or "%s|%s|%s" % (env.get("TORCHELASTIC_RUN_ID", ""), env.get("MASTER_ADDR", ""),
                                                       env.get("MASTER_PORT", "29500"))
    return hashlib.sha256(src.encode("utf-8", "replace")).digest()[:_NONCE_BYTES]


def shard_range(n_items, rank, world_size):
    """Contiguous [begin, end) of rank `rank`; sizes differ by at most one, order preserved."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size: {}/{}".format(rank, world_size))
    base, extra = divmod(n_items, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def rank_env(environ=None):
    """(rank, local_rank, world_size) from the launcher's environment (RANK / LOCAL_RANK /
    WORLD_SIZE as set by ``torch.distributed.run`` or by :func:`launch`); (0, 0, 1) without one."""
    env = os.environ if environ is None else environ
    rank = int(env.get("RANK", "0"))
    return rank, int(env.get("LOCAL_RANK", str(rank))), int(env.get("WORLD_SIZE", "1"))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _candidate_ports(environ):
    if environ.get("QCAT_RDZV_PORT"):
        return [int(environ["QCAT_RDZV_PORT"])]
    base = int(environ.get("MASTER_PORT", "29500"))
    # the launcher's own store listens on MASTER_PORT itself; the ports after it are tried in order
    return [1024 + (base + 1 + i - 1024) % (65536 - 1024) for i in range(_N_CANDIDATE_PORTS)]


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        buf += chunk
    return buf


def exchange_id(rank, world_size, make_id, environ=None, timeout=300.0):
    """Rank 0 calls ``make_id()`` (-> bytes) and serves the result to the other ``world_size - 1``
    ranks over TCP on MASTER_ADDR; every rank returns the same bytes.  Request and reply carry the job's
    nonce (:func:`job_nonce`) and the world size, so a server or a rank of another job on a neighbouring
    port is told apart and skipped, and a rank is counted once."""
    env = os.environ if environ is None else environ
    if world_size == 1:
        return make_id()
    addr = env.get("MASTER_ADDR", "127.0.0.1")
    ports = _candidate_ports(env)
    nonce = job_nonce(env)
    deadline = time.time() + timeout
    if rank == 0:
        srv = None
        for p in ports:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr, p))
                s.listen(world_size)
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise RuntimeError("rendezvous: no free port among {}".format(ports))
        payload = make_id()
        served = set()
        try:
            while len(served) < world_size - 1:
                srv.settimeout(max(0.1, deadline - time.time()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise RuntimeError("rendezvous: only {} of {} ranks arrived".format(len(served) + 1, world_size))
                with conn:
                    conn.settimeout(10.0)
                    try:
                        req = _recv_exact(conn, len(_MAGIC_REQ) + _NONCE_BYTES + 8)
                        w, r = struct.unpack("<ii", req[len(_MAGIC_REQ) + _NONCE_BYTES:])
                        # another job's rank (wrong nonce) or a malformed request gets no id and is not counted
                        if (req[:len(_MAGIC_REQ)] != _MAGIC_REQ or req[len(_MAGIC_REQ):len(_MAGIC_REQ) + _NONCE_BYTES] != nonce
                                or w != world_size or not (0 < r < world_size)):
                            continue
                        # a rank counts once (the set) and only when it has acknowledged the id: one whose read of the
                        # reply failed (socket timeout, reset) asks again and is served again instead of being skipped
                        # until the deadline
                        conn.sendall(_MAGIC_REP + nonce + struct.pack("<i", len(payload)) + payload)
                        if _recv_exact(conn, 1) == _ACK:
                            served.add(r)
                    except (OSError, ConnectionError, struct.error):
                        continue
        finally:
            srv.close()
        return payload
    req = _MAGIC_REQ + nonce + struct.pack("<ii", world_size, rank)
    while time.time() < deadline:
        for p in ports:
            try:
                with socket.create_connection((addr, p), timeout=2.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(req)
                    head = _recv_exact(conn, len(_MAGIC_REP) + _NONCE_BYTES + 4)
                    if head[:len(_MAGIC_REP)] != _MAGIC_REP or head[len(_MAGIC_REP):len(_MAGIC_REP) + _NONCE_BYTES] != nonce:
                        continue                             # some other job's server: try the next port
                    (n,) = struct.unpack("<i", head[len(_MAGIC_REP) + _NONCE_BYTES:])
                    got = _recv_exact(conn, n)
                    conn.sendall(_ACK)
                    return got
            except (OSError, ConnectionError, struct.error):
                continue
        time.sleep(0.05)
    raise RuntimeError("rendezvous: rank {} could not reach rank 0 at {}:{}".format(rank, addr, ports))


COMM_STAGES = ("rendezvous", "ncclCommInitRank", "first all-reduce", "ready")


def init_comm(ctx, rank=None, world_size=None, environ=None, trace=None):
    """The RCCL communicator of this rank: rank 0 creates the unique id in the native library, the
    id travels by :func:`exchange_id`, every rank joins (collective), and one all-reduce of a 1 per rank
    proves the ring before any timed work.  ``trace(stage)`` is called on entering each of
    :data:`COMM_STAGES`; a failure is re-raised naming the rank and the stage it was in -- the three
    places a multi-GPU start-up dies in look alike from outside (a hang or a bare RCCL error code)."""
    r, _lr, w = rank_env(environ)
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    stage = [COMM_STAGES[0]]

    def enter(name):
        stage[0] = name
        if trace is not None:
            trace(name)

    comm = None
    try:
        enter("rendezvous")
        uid = exchange_id(rank, world_size, native.comm_unique_id, environ)
        # RCCL writes a version banner to the C stdout of rank 0 while the communicator is built; callers
        # that print machine-readable results on stdout get it on stderr instead
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            enter("ncclCommInitRank")
            comm = native.NativeComm(ctx, world_size, rank, uid)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        enter("first all-reduce")
        got = comm.allreduce([1.0], native.REDUCE_SUM)[0]
        if int(round(got)) != world_size:
            raise RuntimeError("all-reduce of one per rank gave {} for {} ranks".format(got, world_size))
        enter("ready")
        return comm
    except Exception as e:
        #  a communicator that failed its proving all-reduce is closed, not leaked; the exception keeps its TYPE
        # (callers catch TimeoutError / OSError of the rendezvous as such) and gains the rank and the stage it died in
        if comm is not None:
            try:
                comm.close()
            except Exception:                                            # noqa: BLE001 -- the first failure is the one to report
                pass
        note = "rank {} of {} failed in stage '{}'".format(rank, world_size, stage[0])
        try:
            wrapped = type(e)("{}: {}".format(note, e))
        except Exception:                                                # noqa: BLE001 -- an exception type with another signature
            wrapped = RuntimeError("{}: {}".format(note, e))
        raise wrapped from e


def _cgroup_cpu_quota():
    """CPUs the container may use according to its cgroup quota (None: unlimited)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2
            q, per = fh.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
    except (IOError, OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            q, per = float(fq.read()), float(fp.read())
            return q / per if q > 0 else None
    except (IOError, OSError, ValueError):
        return None


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def rank_cpu_plan(n_ranks, numa_nodes=None, affinity=None, quota=None, node_cpus=None):
    """Host-side placement of the ranks of one node: (cpu set, host threads) per rank.

    A rank's host work is the window-compaction pool of ``qcat_scan_batch`` (up to 16 threads): left alone, 8 ranks
    start 8 x 16 threads on whatever CPUs the scheduler picks -- far from their GPUs and, under the container's CPU
    quota, 8x oversubscribed.  Rank r gets the CPUs of its GPU's NUMA node (``numa_nodes[r]``, -1 / None: unknown ->
    the whole affinity mask), split evenly among the ranks that share the node, and
    ``max(1, usable CPUs // n_ranks)`` host threads where usable = min(affinity mask, cgroup quota)."""
    if affinity is None:
        try:
            affinity = set(os.sched_getaffinity(0))
        except AttributeError:
            affinity = set(range(os.cpu_count() or 1))
    affinity = set(affinity)
    if quota is None:
        quota = _cgroup_cpu_quota()
    usable = len(affinity) if not quota else max(1, min(len(affinity), int(quota + 0.5)))
    threads = max(1, usable // n_ranks)
    numa_nodes = list(numa_nodes) if numa_nodes is not None else [-1] * n_ranks

    def cpus_of(node):
        if node is None or node < 0:
            return None
        if node_cpus is not None:
            return set(node_cpus.get(node, ())) & affinity or None
        try:
            with open("/sys/devices/system/node/node%d/cpulist" % node) as fh:
                return (_parse_cpulist(fh.read()) & affinity) or None
        except (IOError, OSError, ValueError):
            return None

    plan = []
    for r in range(n_ranks):
        local = cpus_of(numa_nodes[r])
        if local is None:
            plan.append((sorted(affinity), threads))
            continue
        mates = [q for q in range(n_ranks) if numa_nodes[q] == numa_nodes[r]]      # ranks sharing the node
        share = sorted(local)
        k, m = mates.index(r), len(mates)
        mine = share[k * len(share) // m:(k + 1) * len(share) // m] or share
        plan.append((mine, threads))
    return plan


def launch(n_ranks, argv, environ=None):
    """Start ``n_ranks`` copies of ``python argv...`` on this node, rank r bound to GPU r through
    RANK / LOCAL_RANK / WORLD_SIZE (the variables ``torch.distributed.run`` sets), to the CPUs of that
    GPU's NUMA node and to its share of the container's CPU quota (:func:`rank_cpu_plan`;
    QCAT_HOST_THREADS tells the native library), and wait for all of them.  Returns the largest exit
    status; a failing rank takes the others down."""
    env = dict(os.environ if environ is None else environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["QCAT_RDZV_PORT"] = str(free_port())
    env["QCAT_RDZV_NONCE"] = uuid.uuid4().hex             # this job's rendezvous only (exchange_id)
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n_ranks)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        lib = native.HipLibrary.get().lib
        nodes = [lib.qcat_device_numa_node(r) for r in range(n_ranks)]
    except (RuntimeError, OSError, AttributeError):
        nodes = None
    plan = rank_cpu_plan(n_ranks, nodes)
    procs = []
    for r in range(n_ranks):
        e = dict(env)
        e["RANK"] = e["LOCAL_RANK"] = str(r)
        cpus, threads = plan[r]
        e.setdefault("QCAT_HOST_THREADS", str(threads))

        def bind(cpus=cpus):
            try:
                os.sched_setaffinity(0, cpus)
            except (AttributeError, OSError, ValueError):
                pass
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, preexec_fn=bind))
    worst = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = max(worst, abs(rc) or 1)
                for q in alive:                      # one rank failed: the collective can never complete
                    q.terminate()
        time.sleep(0.05)
    return worst


# ---- ONE file over N ranks (round 6) ---------------------------------------------------------------------------------------
# The reference driver is one process and one file (qcat/cli.py:445-563).  Its unit is the BATCH of 4000 reads: the kit vote and
# --filter-barcodes are decided per batch (cli.py:500-513, scanner_base.py:690-733; SURVEY.md 8e "what does not shard"), the
# reads of a batch are independent.  So a file shards at batch boundaries: every rank learns where the batches start
# (qcat_fastq_batch_offsets: one pass of the record splitter, no device), takes a contiguous run of WHOLE batches, demultiplexes
# its byte range with outputs of its own (qcat_fastq_demux_stream, range_begin / range_end), and the ranks' outputs strung
# together in rank order ARE the one-rank outputs, byte for byte; the histograms meet in one all-reduce.

def file_shards(path, world_size, batch_size=4000, segment_bytes=0):
    """[(begin, end)] byte ranges of `path`, one per rank (empty ranges for ranks beyond the batches), plus (n_reads, next_offset):
    whole batches of `batch_size` reads, contiguous, in rank order.  next_offset < file size: the plain records end there and the
    caller's own parser takes the rest (after the last shard)."""
    from . import native
    offs, n_reads, next_offset = native.FastqFile.batch_offsets(path, batch_size, segment_bytes)
    n_batches = len(offs) - 1
    shards = []
    for r in range(world_size):
        b0, b1 = shard_range(n_batches, r, world_size)
        shards.append((int(offs[b0]), int(offs[b1])))
    return shards, n_reads, next_offset


def shard_paths(path, rank):
    """where rank `rank` writes its part of the output `path` (a TSV / FASTQ file or a per-barcode directory)"""
    return "%s.rank%d" % (path, rank)


def demux_file_shard(detector, reads_fq, rank, world_size, qcat_config, tsv_path=None, out_dir=None, out_path=None, trim=False,
                     min_read_length=0, nobatch=False, filter_barcodes=False, batch_size=4000, comm=None, shards=None):
    """This rank's share of ONE file: its whole batches through the native file loop, outputs to `shard_paths(...)`, histograms
    summed over the ranks when a communicator is given (any object with allreduce(values, op) -- native.NativeComm, a stub).
    Returns (barcode counts, adapter counts, n_none, n_adapter_none, stats) with GLOBAL counts when `comm` is given.  The caller
    runs `merge_shards` on one rank afterwards."""
    import numpy as np
    from . import native
    layouts = detector.layouts
    if shards is None:
        shards, _, _ = file_shards(reads_fq, world_size, batch_size)
    begin, end = shards[rank]
    one_kit = len(set(l.kit for l in layouts)) == 1
    kit_auto = (not nobatch) and not one_kit
    kit = detector._native_kit(layouts, qcat_config, native.ENDS_BOTH)
    dual = detector._native_mode == "dual"
    sinks = []

    def fd_of(p):
        fh = open(shard_paths(p, rank), "wb")
        sinks.append(fh)
        return fh.fileno()
    my_dir = None
    if out_dir:
        my_dir = shard_paths(out_dir, rank)
        os.makedirs(my_dir, exist_ok=True)
    try:
        if end > begin:
            bc, ad, n_none, n_ad_none, stats = native.FastqFile.demux_stream(
                reads_fq, detector._context(), kit, layouts, dual, batch_size=batch_size, kit_auto=kit_auto, trim=trim,
                min_read_length=min_read_length, tsv_fd=fd_of(tsv_path) if tsv_path else None,
                out_fd=fd_of(out_path) if (out_path and not out_dir and not tsv_path) else None, out_dir=my_dir,
                filter_barcodes=bool(filter_barcodes) and not nobatch, byte_range=(begin, end))
        else:                                        # more ranks than batches: an empty shard (its outputs exist and are empty)
            n_t = len(layouts)
            w0 = max(1, max(len(l.get_barcode_set(0) or ()) for l in layouts))
            w1 = max(1, max(len(l.get_barcode_set(1) or ()) for l in layouts)) if dual else 1
            bc, ad, n_none, n_ad_none = np.zeros((n_t, w0, w1), dtype=np.int64), np.zeros(n_t, dtype=np.int64), 0, 0
            stats = {"n_reads": 0, "n_skipped": 0, "file_bytes": 0, "parse_s": 0.0, "scan_s": 0.0, "write_s": 0.0, "total_s": 0.0,
                     "next_offset": begin, "incomplete": 0, "segments": 0}
            if tsv_path:
                fd_of(tsv_path)
            elif out_path and not out_dir:
                fd_of(out_path)
    finally:
        for fh in sinks:
            fh.close()
    if comm is not None and world_size > 1:
        flat = np.concatenate([bc.reshape(-1), ad.reshape(-1), [n_none, n_ad_none, stats["n_reads"], stats["n_skipped"]]]).astype(np.float64)
        tot = np.asarray(comm.allreduce(flat.tolist(), native.REDUCE_SUM), dtype=np.float64)       # (exact: counts below 2^53)
        nb, na = bc.size, ad.size
        bc = np.rint(tot[:nb]).astype(np.int64).reshape(bc.shape)
        ad = np.rint(tot[nb:nb + na]).astype(np.int64)
        n_none, n_ad_none = int(round(tot[nb + na])), int(round(tot[nb + na + 1]))
        stats = dict(stats, n_reads_total=int(round(tot[nb + na + 2])), n_skipped_total=int(round(tot[nb + na + 3])))
    return bc, ad, n_none, n_ad_none, stats


def merge_shards(path, world_size, is_dir=False, keep=False):
    """string the ranks' shards of `path` together in rank order (one rank calls this after all are done: the result is the
    one-rank output byte for byte).  Directories: per file name, in rank order.  The bytes move inside the kernel
    (copy_file_range where the file system has it)."""
    import shutil
    if is_dir:
        os.makedirs(path, exist_ok=True)
        names = []
        for r in range(world_size):
            d = shard_paths(path, r)
            if os.path.isdir(d):
                for n in sorted(os.listdir(d)):
                    if n not in names:
                        names.append(n)
        for n in names:
            with open(os.path.join(path, n), "wb") as dst:
                for r in range(world_size):
                    src_path = os.path.join(shard_paths(path, r), n)
                    if os.path.exists(src_path):
                        with open(src_path, "rb") as src:
                            shutil.copyfileobj(src, dst, 1 << 24)
        if not keep:
            for r in range(world_size):
                shutil.rmtree(shard_paths(path, r), ignore_errors=True)
        return
    with open(path, "wb") as dst:
        for r in range(world_size):
            sp = shard_paths(path, r)
            if os.path.exists(sp):
                with open(sp, "rb") as src:
                    shutil.copyfileobj(src, dst, 1 << 24)
                if not keep:
                    os.unlink(sp)
