"""The native FASTQ record splitter (qcat_fastq_open, csrc/fastq_host.inc) on the CPU: it must index plain four-line ASCII
FASTQ files exactly as the Python parser of the driver (which restates Biopython's FastqGeneralIterator) reads them, and
answer "unsupported" -- so that the Python parser takes the file -- for everything that parser would read differently."""
import os
import random

import pytest

import helpers
from qcat_amd import cli, native


def _index(path):
    fq = native.FastqFile(path)
    try:
        with open(path, "rb") as fh:
            data = fh.read()
        out = []
        for r in range(fq.n_reads):
            to, tl, so, sl = fq.read_info(r)
            out.append((data[to:to + tl].decode(), data[so:so + sl].decode()))
        return out
    finally:
        fq.close()


def _python_records(path):
    with open(path) as fh:
        return [(t, s) for t, s, _q in cli._fastq_records(fh)]


@pytest.mark.parametrize("name", ["nbd103.fastq", "pbk004.fastq", "rab204.fastq", "rbk004.fastq"])
def test_shipped_files_index_like_the_python_parser(name):
    path = os.path.join(helpers.GOLDEN, "data", name)
    assert _index(path) == _python_records(path)


def test_parallel_split_of_a_file_with_awkward_quality_lines(tmp_path, monkeypatch):
    """Quality lines that begin with '@' or '+', titles with tabs and trailing blanks, a '+' line repeating the title, a last
    record without a newline -- in a file big enough to be split over several threads (QCAT_HOST_THREADS)."""
    rng = random.Random(7)
    monkeypatch.setenv("QCAT_HOST_THREADS", "5")
    lines = []
    for i in range(60000):
        n = rng.randrange(1, 400)
        seq = "".join(rng.choice("ACGTN") for _ in range(n))
        qual = "".join(rng.choice("@+!I5#") for _ in range(n))
        title = "read%d" % i + rng.choice(["", " ch=1", "\tcomment with\ttabs", " trailing  "])
        plus = "+" + (title.rstrip() if i % 7 == 0 else "")
        lines += ["@" + title, seq, plus, qual]
    path = str(tmp_path / "awkward.fastq")
    with open(path, "w") as fh:
        fh.write("\n".join(lines))                       # (no newline after the last record)
    assert os.path.getsize(path) > 5 * (4 << 20)         # more than one piece
    got, want = _index(path), _python_records(path)
    assert len(got) == 60000 and got == want


@pytest.mark.parametrize("body", [
    "@r1\nACGT\n+\nIIII\n\n@r2\nAC\n+\nII\n",           # blank line between records
    "@r1\nAC\nGT\n+\nIIII\n",                           # wrapped sequence
    "@r1\r\nACGT\r\n+\r\nIIII\r\n",                     # CRLF
    "@r1\nACGT \n+\nIIII\n",                            # trailing blank on the sequence line
    "@r1\nACGT\n+r2\nIIII\n",                           # '+' line with another title
    "@r1\nACGT\n+\nIII\n",                              # quality shorter than the sequence
    ">r1\nAC\nGT\n",                                    # FASTA with a wrapped sequence
    ">r1\nACGT\n\n>r2\nAC\n",                          # FASTA with a blank line
    ">r1\n>r2\nAC\n",                                   # FASTA record without a sequence
    "; comment\n>r1\nACGT\n",                           # something before the first record
    "@r1\nAC\xc3\xa9T\n+\nIIII\n",                      # non-ASCII
    "@r1\nACGT\n+\n",                                   # truncated
])
def test_anything_else_is_left_to_the_python_parser(body, tmp_path):
    path = str(tmp_path / "odd.fastq")
    with open(path, "w", encoding="latin-1") as fh:
        fh.write(body)
    with pytest.raises(native.FastqFile.Unsupported):
        native.FastqFile(path)


def test_plain_two_line_fasta_indexes_like_the_python_parser(tmp_path, monkeypatch):
    """Round 4: plain FASTA ('>' title, one sequence line per read) on the native path too -- same records as the Python
    parser (Biopython's SimpleFastaParser rules, cli._fasta_records), split over several threads, last record without a
    newline."""
    rng = random.Random(11)
    monkeypatch.setenv("QCAT_HOST_THREADS", "4")
    lines = []
    for i in range(50000):
        seq = "".join(rng.choice("ACGTN") for _ in range(rng.randrange(1, 900)))
        lines += [">read%d" % i + rng.choice(["", " ch=1", "\tcomment with\ttabs", " trailing  "]), seq]
    path = str(tmp_path / "plain.fasta")
    with open(path, "w") as fh:
        fh.write("\n".join(lines))
    assert os.path.getsize(path) > 4 * (4 << 20)
    with open(path) as fh:
        want = [(t, s) for t, s in cli._fasta_records(fh)]
    got = _index(path)
    assert len(got) == 50000 and got == want


def test_empty_file_and_missing_file(tmp_path):
    path = str(tmp_path / "empty.fastq")
    open(path, "w").close()
    fq = native.FastqFile(path)
    assert fq.n_reads == 0
    fq.close()
    with pytest.raises(RuntimeError):
        native.FastqFile(str(tmp_path / "nope.fastq"))


# ---- the reader stage of qcat_fastq_demux_stream (csrc/fastq_stream.inc): segments, carries, batch cuts -----------------------

def _awkward_fastq(path, n, seed, newline_at_end=True):
    rng = random.Random(seed)
    lines, bases = [], 0
    for i in range(n):
        k = rng.randrange(1, 400)
        seq = "".join(rng.choice("ACGTN") for _ in range(k))
        qual = "".join(rng.choice("@+!I5#") for _ in range(k))
        title = "read%d" % i + rng.choice(["", " ch=1", "\tcomment with\ttabs", " trailing  "])
        lines += ["@" + title, seq, "+" + (title.rstrip() if i % 7 == 0 else ""), qual]
        bases += k
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + ("\n" if newline_at_end else ""))
    return bases


@pytest.mark.parametrize("segment_bytes,batch", [(64, 0), (1000, 0), (1000, 7), (4096, 100), (100000, 4000), (1 << 20, 0), (0, 4000)])
def test_streamed_segments_see_every_read_once(segment_bytes, batch, tmp_path, monkeypatch):
    """Segments far smaller than a record, than a batch, or bigger than the file; quality lines that start with '@' / '+';
    with and without a final newline: the reads and letters of all rounds add up to the file's, whatever the cut."""
    monkeypatch.setenv("QCAT_HOST_THREADS", "3")
    for nl in (True, False):
        path = str(tmp_path / ("awk%d.fastq" % nl))
        bases = _awkward_fastq(path, 3000, 5, newline_at_end=nl)
        n, nb, off, segs = native.FastqFile.stream_count(path, segment_bytes, batch)
        assert (n, nb, off) == (3000, bases, os.path.getsize(path))
        assert native.FastqFile.stream_count(path, segment_bytes, batch, reader=2) == (n, nb, off, segs)     # mapped windows instead of pread
        # (a batch that holds the whole file makes ONE round of it, however small the segments)
        assert segs >= 1 and (segment_bytes == 0 or segment_bytes >= (1 << 20) or batch >= 3000 or segs > 1)


def test_streamed_segments_of_a_big_file_split_on_several_threads(tmp_path, monkeypatch):
    monkeypatch.setenv("QCAT_HOST_THREADS", "5")
    path = str(tmp_path / "big.fastq")
    bases = _awkward_fastq(path, 120000, 9)
    assert os.path.getsize(path) > 40 << 20
    for seg, batch in ((16 << 20, 4000), (7 << 20, 0)):
        for reader in (1, 2):
            n, nb, off, segs = native.FastqFile.stream_count(path, seg, batch, reader=reader)
            assert (n, nb, off) == (120000, bases, os.path.getsize(path)) and segs >= 3


def test_streamed_fasta_and_the_hand_back_at_a_record_that_is_not_plain(tmp_path):
    rng = random.Random(3)
    recs = [(">r%d c=%d" % (i, i), "".join(rng.choice("ACGT") for _ in range(rng.randrange(1, 300)))) for i in range(2000)]
    path = str(tmp_path / "plain.fasta")
    with open(path, "w") as fh:
        fh.write("".join(t + "\n" + s + "\n" for t, s in recs))
    n, nb, off, _ = native.FastqFile.stream_count(path, 5000, 50)
    assert (n, nb, off) == (2000, sum(len(s) for _t, s in recs), os.path.getsize(path))
    # round 6: a WRAPPED record in the middle is taken (its segment is rewritten as plain records, tests/test_fastq_wrapped.py)
    text = "".join("@q%d\n%s\n+\n%s\n" % (i, "ACGT" * 20, "I" * 80) for i in range(1000))
    tail = "".join("@z%d\nAC\n+\nII\n" % i for i in range(10))
    path = str(tmp_path / "wrapped_later.fastq")
    with open(path, "w") as fh:
        fh.write(text + "@wrapped\nACGT\nACGT\n+\nIIIIIIII\n" + tail)
    n, nb, off, _ = native.FastqFile.stream_count(path, 10000, 10)
    assert (n, nb, off) == (1011, 80 * 1000 + 8 + 20, os.path.getsize(path))
    # a record the Python parser has to report itself (the captions differ) in the middle: the rounds end in front of ITS
    # segment, at a batch boundary, and say where
    bad = text + "@odd\nACGT\n+other\nIIII\n" + tail
    path = str(tmp_path / "odd_later.fastq")
    with open(path, "w") as fh:
        fh.write(bad)
    n, _nb, off, _ = native.FastqFile.stream_count(path, 10000, 10)
    assert 0 < n < 1000 and n % 10 == 0 and bad[off:off + 2] == "@q" and bad[:off].count("\n") == 4 * n
    # ... and in the very first segment nothing is handed out at all
    path = str(tmp_path / "odd_first.fastq")
    with open(path, "w") as fh:
        fh.write("@odd\nACGT\n+other\nIIII\n" + text)
    with pytest.raises(native.FastqFile.Unsupported):
        native.FastqFile.stream_count(path, 10000, 10)
    empty = str(tmp_path / "empty.fastq")
    open(empty, "w").close()
    assert native.FastqFile.stream_count(empty)[:2] == (0, 0)
