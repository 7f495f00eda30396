"""N > 1 path on CPU: world_size-2 gloo processes shard a batch by read range, each rank produces
the count vector of its shard (here with the CPU oracle -- the GPU kernels are covered by the
-m gpu tests), and the all-reduced vector must equal the counts of the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
import synth
from qcat_amd import native, parallel, scanner


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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_reads, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    det = scanner.factory(kit="PBC096")
    desc = det.descriptor()
    b, e = parallel.shard_range(n_reads, rank, world)
    reads = synth.synth_batch(e - b, 4711, det.layouts, 1, 0, first=b, error_rate=0.08)
    _, cnt = oracle_lib.scan(desc, reads, counts=True)
    total = parallel.allreduce_counts(cnt, dist)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), total)
    dist.barrier()
    dist.destroy_process_group()


def test_count_allreduce_world2(tmp_path):
    n = 240
    mp.spawn(_worker, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    det = scanner.factory(kit="PBC096")
    reads = synth.synth_batch(n, 4711, det.layouts, 1, 0, error_rate=0.08)
    _, want = oracle_lib.scan(det.descriptor(), reads, counts=True)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, want)
    assert want[:96].sum() + want[96] == n          # every read lands in exactly one barcode bucket
