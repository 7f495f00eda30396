#!/usr/bin/env python3
"""Rehearsal of `bench.py --gpus N` on a box with FEWER than N devices (task 8): test infrastructure only.

    python tests/rehearse_gpus8.py --launch --gpus 8 --reads 1000000 --steps 2 --warmup 1 --cpu-seconds 1

`--launch` starts the N ranks through the product's own launcher (qcat_amd.parallel.launch: RANK / LOCAL_RANK / WORLD_SIZE,
NUMA / CPU-quota placement, QCAT_HOST_THREADS) with this file as the rank program; a rank then runs the UNMODIFIED
`bench.main()` with three test-only substitutions, none of them inside the library or the package:
  * every rank's context lives on device 0 (the box has one GPU) and the device count is reported as N;
  * the communicator is a stub over one TCP socket per rank (rank 0 sums / maximises and answers): the count vectors are
    fetched from the device, summed on rank 0, written back with hipMemcpy -- RCCL refuses several ranks on one device;
  * nothing else: seeds, shards, the timed loop, the barrier + max-over-ranks, host_inclusive at `usable CPUs / N` host
    threads per rank, cpu_baseline and parity on rank 0 and the ONE JSON line are bench.py's.
The throughput it prints is that of N processes sharing ONE GPU -- not a scaling figure; what the rehearsal proves is that
the N-rank code path runs to its final line before the first real 8-GPU run."""
import ctypes as C
import os
import socket
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from qcat_amd import native, parallel  # noqa: E402


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rehearsal peer closed")
        buf += chunk
    return buf


class TcpComm(object):
    """what bench.py asks of native.NativeComm, over TCP: rank 0 reduces and answers"""

    def __init__(self, ctx, rank, world):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.hip = native.HipLibrary.get()
        port = int(os.environ["QCAT_REHEARSAL_PORT"])
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", port))
            srv.listen(world)
            self.peers = {}
            while len(self.peers) < world - 1:
                conn, _ = srv.accept()
                (r,) = struct.unpack("<i", _recv_exact(conn, 4))
                self.peers[r] = conn
            srv.close()
        else:
            import time
            deadline = time.time() + 120
            while True:
                try:
                    self.conn = socket.create_connection(("127.0.0.1", port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.1)
            self.conn.settimeout(600.0)
            self.conn.sendall(struct.pack("<i", rank))
        self._hipMemcpy = None

    def _reduce(self, arr, op):
        """arr: numpy array (float64 or int64); returns the reduction over all ranks (same dtype)"""
        raw = arr.tobytes()
        if self.rank == 0:
            acc = arr.copy()
            for r in sorted(self.peers):
                other = np.frombuffer(_recv_exact(self.peers[r], len(raw)), dtype=arr.dtype)
                acc = np.maximum(acc, other) if op == native.REDUCE_MAX else acc + other
            out = acc.tobytes()
            for r in sorted(self.peers):
                self.peers[r].sendall(out)
            return acc
        self.conn.sendall(raw)
        return np.frombuffer(_recv_exact(self.conn, len(raw)), dtype=arr.dtype).copy()

    def allreduce(self, values, op=native.REDUCE_SUM):
        return list(self._reduce(np.asarray([float(v) for v in values], dtype=np.float64), op))

    def barrier(self):
        self.hip.check(self.hip.lib.qcat_ctx_synchronize(self.ctx.handle))     # (qcat_comm_barrier drains the stream first too)
        self._reduce(np.zeros(1, dtype=np.float64), native.REDUCE_SUM)

    def allreduce_counts(self):
        lib = self.hip.lib
        n = int(self.n_buckets)
        host = np.zeros(n, dtype=np.int64)
        self.hip.check(lib.qcat_ctx_fetch_counts(self.ctx.handle, host.ctypes.data, n))
        total = self._reduce(host, native.REDUCE_SUM)
        if self._hipMemcpy is None:
            rt = C.CDLL("libamdhip64.so")
            rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            self._hipMemcpy = rt.hipMemcpy
        lib.qcat_ctx_counts_devptr.restype = C.c_void_p
        dev = lib.qcat_ctx_counts_devptr(self.ctx.handle)
        if self._hipMemcpy(dev, total.ctypes.data, n * 8, 1) != 0:       # hipMemcpyHostToDevice
            raise RuntimeError("rehearsal: hipMemcpy of the count vector failed")

    def close(self):
        pass


def rank_main():
    import bench
    rank, local_rank, world = parallel.rank_env()
    hip = native.HipLibrary.get()
    real_count = hip.lib.qcat_device_count()
    if real_count < 1:
        sys.exit("rehearsal: no HIP device")
    hip.lib.qcat_device_count = lambda: max(real_count, world)          # test-only: bench.py's device checks pass
    real_ctx = native.NativeContext

    class Ctx0(real_ctx):
        def __init__(self, device=0):
            real_ctx.__init__(self, 0)                                   # every rank on device 0
    native.NativeContext = Ctx0
    real_make = bench.make_scanner
    bench.make_scanner = lambda workload, mode, kit_name, device: real_make(workload, mode, kit_name, 0)

    def init_comm(ctx, r=None, w=None, environ=None, trace=None):
        for stage in parallel.COMM_STAGES[:2]:
            if trace is not None:
                trace(stage)
        comm = TcpComm(ctx, rank if r is None else r, world if w is None else w)
        for stage in parallel.COMM_STAGES[2:]:
            if trace is not None:
                trace(stage)
        return comm
    parallel.init_comm = init_comm
    # the stub needs the bucket count of the kit bench.py builds
    real_kit = native.NativeKit

    class Kit(real_kit):
        def __init__(self, descriptor, jit=None):
            real_kit.__init__(self, descriptor, jit=jit)
            TcpComm.n_buckets = descriptor.n_count_buckets
    native.NativeKit = Kit
    sys.argv = [os.path.join(ROOT, "bench.py")] + [a for a in sys.argv[1:] if a != "--launch"]
    bench.main()


def main():
    if "--launch" in sys.argv:
        n = int(sys.argv[sys.argv.index("--gpus") + 1])
        env = dict(os.environ)
        env["QCAT_REHEARSAL_PORT"] = str(parallel.free_port())
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        sys.exit(parallel.launch(n, [os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--launch"], environ=env))
    rank_main()


if __name__ == "__main__":
    main()
