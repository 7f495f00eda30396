"""ONE file over N ranks (round 6, qcat_amd/parallel.py: file_shards / demux_file_shard / merge_shards).

CPU: where the batches of the driver's loop start in a file (qcat_fastq_batch_offsets: the record splitter alone, no device),
the contiguous whole-batch shards, the merge of the ranks' outputs.
GPU (-m gpu): two ranks -- two host threads, each with its own context on device 0, a stub communicator -- demultiplex one
kit-auto file with --filter-barcodes; their outputs strung together in rank order are the one-rank run's outputs byte for byte
and the summed histograms are its histograms (vote and filter are per batch of 4000 reads, qcat/cli.py:500-513: SURVEY.md 8e)."""
import hashlib
import os
import threading

import numpy as np
import pytest

import helpers  # noqa: F401
import synth
from qcat_amd import config, native, parallel, scanner


def _write_fastq(path, n, reads, fasta=False, comment=True):
    with open(path, "w") as fh:
        for i in range(n):
            r = reads[i % len(reads)]
            title = "r%d ch=%d" % (i, i % 512) if comment else "r%d" % i
            if fasta:
                fh.write(">%s\n%s\n" % (title, r))
            else:
                fh.write("@%s\n%s\n+\n%s\n" % (title, r, "I" * len(r)))


@pytest.fixture(scope="module")
def reads():
    det = scanner.factory(kit="PBC096")
    return synth.synth_batch(200, 5, det.layouts, 1, 0, error_rate=0.05)


@pytest.mark.parametrize("fasta", [False, True])
@pytest.mark.parametrize("n,bs", [(0, 4000), (1, 4000), (3999, 4000), (4000, 4000), (4001, 4000), (10123, 4000), (10123, 37)])
def test_batch_offsets(tmp_path, reads, n, bs, fasta):
    p = str(tmp_path / ("r.fasta" if fasta else "r.fastq"))
    _write_fastq(p, n, reads, fasta=fasta)
    data = open(p, "rb").read()
    marker = b">" if fasta else b"@"
    for seg in (0, 1 << 20):                                            # one segment / many small ones: the same answer
        offs, n_reads, nxt = native.FastqFile.batch_offsets(p, bs, segment_bytes=seg)
        assert n_reads == n and nxt == len(data) and offs[-1] == len(data)
        assert len(offs) - 1 == (n + bs - 1) // bs
        for i, o in enumerate(offs[:-1]):
            assert data[o:o + 1] == marker and data[o + 1:o + 24].split(b" ")[0] == b"r%d" % (i * bs)


def test_batch_offsets_stop_in_front_of_a_record_that_is_not_plain(tmp_path, reads):
    p = str(tmp_path / "r.fastq")
    _write_fastq(p, 9000, reads)
    good = os.path.getsize(p)
    with open(p, "a") as fh:                                            # a wrapped record, then plain ones again
        fh.write("@wrapped\n%s\n%s\n+\n%s\n" % (reads[0][:50], reads[0][50:], "I" * len(reads[0])))
    offs, n_reads, nxt = native.FastqFile.batch_offsets(p, 4000, segment_bytes=4 << 20)
    # the native loop ends in front of the SEGMENT that holds the odd record: a batch boundary at or before it
    assert nxt <= good and nxt in set(int(o) for o in offs) and n_reads % 4000 == 0 and n_reads <= 9000
    shards, n2, nxt2 = parallel.file_shards(p, 3, 4000, segment_bytes=4 << 20)
    assert (n2, nxt2) == (n_reads, nxt) and shards[0][0] == 0 and shards[-1][1] == nxt
    with pytest.raises(native.FastqFile.Unsupported):
        bad = str(tmp_path / "bad.fastq")
        with open(bad, "w") as fh:
            fh.write("@a\nACGT\nACGT\n+\nIIIIIIII\n")
        native.FastqFile.batch_offsets(bad, 4000)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_are_whole_batches_in_rank_order(tmp_path, reads, world):
    p = str(tmp_path / "r.fastq")
    _write_fastq(p, 21017, reads)
    offs, n_reads, _ = native.FastqFile.batch_offsets(p, 4000)
    shards, n2, nxt = parallel.file_shards(p, world, 4000)
    assert n2 == n_reads == 21017 and len(shards) == world
    assert shards[0][0] == 0 and shards[-1][1] == os.path.getsize(p)
    starts = set(int(o) for o in offs)
    for r, (b, e) in enumerate(shards):
        assert b in starts and e in starts and b <= e
        if r:
            assert b == shards[r - 1][1]                                 # contiguous
    sizes = [sum(1 for o in offs[:-1] if b <= o < e) for b, e in shards]
    assert sum(sizes) == len(offs) - 1 and max(sizes) - min(sizes) <= 1  # batches spread evenly


def test_merge_shards(tmp_path):
    p = str(tmp_path / "calls.tsv")
    for r, text in enumerate((b"a\nb\n", b"", b"c\n")):
        with open(parallel.shard_paths(p, r), "wb") as fh:
            fh.write(text)
    parallel.merge_shards(p, 3)
    assert open(p, "rb").read() == b"a\nb\nc\n" and not os.path.exists(parallel.shard_paths(p, 0))
    d = str(tmp_path / "bc")
    for r, files in enumerate(({"barcode01.fastq": b"1", "none.fastq": b"n0"}, {"barcode02.fastq": b"2", "none.fastq": b"n1"})):
        os.makedirs(parallel.shard_paths(d, r))
        for name, text in files.items():
            with open(os.path.join(parallel.shard_paths(d, r), name), "wb") as fh:
                fh.write(text)
    parallel.merge_shards(d, 2, is_dir=True)
    assert sorted(os.listdir(d)) == ["barcode01.fastq", "barcode02.fastq", "none.fastq"]
    assert open(os.path.join(d, "none.fastq"), "rb").read() == b"n0n1"


class _ThreadComm(object):
    """all-reduce among the threads of one process (the stub of the two-rank test: RCCL refuses several ranks per device)"""

    def __init__(self, n):
        self.n, self.lock, self.barrier = n, threading.Lock(), threading.Barrier(n)
        self.acc = None

    def allreduce(self, values, op=native.REDUCE_SUM):
        v = np.asarray(values, dtype=np.float64)
        with self.lock:
            self.acc = v.copy() if self.acc is None else self.acc + v
        self.barrier.wait()
        out = self.acc.copy()
        self.barrier.wait()
        with self.lock:
            self.acc = None
        self.barrier.wait()
        return out.tolist()


def _sha_dir(d):
    out = {}
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            out[f] = hashlib.sha256(fh.read()).hexdigest()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("form", ["tsv", "dir"])
def test_two_ranks_on_one_file_equal_the_one_rank_run(tmp_path, world, form):
    """kit auto + --filter-barcodes (both per batch) on a file of 5.3 batches: barcode 1 and 2 common, a dozen rare ones the
    filter drops, reads of a second kit in the third batch so that the batches vote differently"""
    det = scanner.factory(mode="epi2me", kit=None)
    cfg = config.qcatConfig()
    nbd = scanner.factory(kit="NBD104/NBD114").layouts
    pbc = scanner.factory(kit="PBC096").layouts
    p = str(tmp_path / "reads.fastq")
    n = 21300
    with open(p, "w") as fh:
        for i in range(n):
            lays = pbc if 8000 <= i < 12000 else nbd
            b = (2 + i // 10 % 12) if i % 10 == 9 else (0 if i % 10 < 6 else 1)
            seq = synth.synth_read(i, 31, lays, 1, 0, error_rate=0.06, insert_len=300, force_barcode=b)
            fh.write("@read%05d runid=x ch=%d\n%s\n+\n%s\n" % (i, 1 + i % 512, seq, "I" * len(seq)))
    kw = dict(qcat_config=cfg, trim=True, min_read_length=100, filter_barcodes=True)
    one = str(tmp_path / "one")
    tsv1, dir1 = one + ".tsv", one + "_bc"
    ref = parallel.demux_file_shard(det, p, 0, 1, tsv_path=tsv1 if form == "tsv" else None, out_dir=dir1 if form == "dir" else None, **kw)
    parallel.merge_shards(tsv1 if form == "tsv" else dir1, 1, is_dir=form == "dir")
    shards, n_reads, _ = parallel.file_shards(p, world)
    assert n_reads == n
    comm = _ThreadComm(world)
    many = str(tmp_path / "many")
    tsvn, dirn = many + ".tsv", many + "_bc"
    got, errors = [None] * world, []

    def rank_main(r):
        try:
            d = scanner.factory(mode="epi2me", kit=None)                 # (its own scanner: the thread's own context on device 0)
            got[r] = parallel.demux_file_shard(d, p, r, world, tsv_path=tsvn if form == "tsv" else None,
                                               out_dir=dirn if form == "dir" else None, comm=comm, shards=shards, **kw)
        except Exception as e:                                           # noqa: BLE001
            errors.append((r, e))
            comm.barrier.abort()
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    parallel.merge_shards(tsvn if form == "tsv" else dirn, world, is_dir=form == "dir")
    if form == "tsv":
        assert open(tsvn, "rb").read() == open(tsv1, "rb").read() and os.path.getsize(tsv1) > 0
    else:
        assert _sha_dir(dirn) == _sha_dir(dir1) and len(os.listdir(dir1)) >= 3
    for r in range(world):                                               # every rank holds the GLOBAL histograms
        assert np.array_equal(got[r][0], ref[0]) and np.array_equal(got[r][1], ref[1])
        assert (got[r][2], got[r][3]) == (ref[2], ref[3])
        assert got[r][4]["n_reads_total"] == ref[4]["n_reads"] and got[r][4]["n_skipped_total"] == ref[4]["n_skipped"]
    assert len(set(l.kit for l in det.layouts)) > 1 and ref[1].sum() > 0
