"""Round 6: files that are not all plain records on the native file loop.  A segment with wrapped sequence / quality lines,
\\r\\n line ends, blank lines or trailing blanks is rewritten as plain records by Biopython's rules (csrc/fastq_stream.inc:
fq_seg_normalize; the reference parses with FastqGeneralIterator / SimpleFastaParser, qcat/cli.py:235-306, restated in
qcat_amd/cli.py).  CPU: the reader stage alone (qcat_fastq_stream_count) against the Python parser -- reads, letters, where the
native loop stops.  GPU: the driver's outputs on such files, native loop against Python loop, byte for byte."""
import io
import os
import random

import pytest

import helpers  # noqa: F401
import synth
from qcat_amd import cli, config, native, scanner


def _reads(n, seed=3):
    det = scanner.factory(kit="PBC096")
    return synth.synth_batch(n, seed, det.layouts, 1, 0, error_rate=0.05, insert_len=200)


def _wrap(s, w):
    return "\n".join(s[i:i + w] for i in range(0, len(s), w))


def _python_counts(path, fastq):
    with open(path) as fh:
        recs = list(cli._fastq_records(fh)) if fastq else [(t, s, None) for t, s in cli._fasta_records(fh)]
    return len(recs), sum(len(r[1]) for r in recs), recs


FORMS = ["fasta60", "fasta_crlf", "fastq80", "fastq_blank_lines", "fastq_plus_title", "fastq_crlf", "fastq_mixed"]


def _write(path, form, reads, rng):
    with open(path, "w", newline="") as fh:
        for i, r in enumerate(reads):
            q = "".join(chr(33 + rng.randrange(60)) for _ in r)
            q = ("@" + q[1:]) if i % 7 == 0 else q                       # quality lines may start with '@'
            t = "read%d runid=w ch=%d" % (i, i % 512)
            if form == "fasta60":
                fh.write(">%s\n%s\n" % (t, _wrap(r, 60)))
            elif form == "fasta_crlf":
                fh.write(">%s  \r\n%s\r\n" % (t, _wrap(r, 70).replace("\n", "\r\n")))
            elif form == "fastq80":
                fh.write("@%s\n%s\n+\n%s\n" % (t, _wrap(r, 80), _wrap(q, 80)))
            elif form == "fastq_blank_lines":
                fh.write("@%s\n%s\n+\n%s\n%s" % (t, r, q, "\n" if i % 3 == 0 else ""))
            elif form == "fastq_plus_title":
                fh.write("@%s\n%s\n+%s\n%s\n" % (t, r, t if i % 2 else "", q))
            elif form == "fastq_crlf":
                fh.write("@%s\r\n%s\r\n+\r\n%s\r\n" % (t, r, q))
            else:                                                        # plain records first, wrapped ones later
                fh.write("@%s\n%s\n+\n%s\n" % (t, r if i < len(reads) // 2 else _wrap(r, 100), q if i < len(reads) // 2 else _wrap(q, 100)))


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("seg", [0, 1 << 16])
def test_reader_takes_what_biopython_takes(tmp_path, form, seg):
    rng = random.Random(5)
    reads = _reads(700)
    p = str(tmp_path / ("r." + ("fasta" if form.startswith("fasta") else "fastq")))
    _write(p, form, reads, rng)
    n, bases, recs = _python_counts(p, not form.startswith("fasta"))
    assert n == len(reads) and [r[1] for r in recs] == reads
    for bs in (0, 64):
        got = native.FastqFile.stream_count(p, segment_bytes=seg, batch_size=bs)
        assert got[0] == n and got[1] == bases and got[2] == os.path.getsize(p), (form, seg, bs, got)


def test_what_the_python_parser_must_report_itself_is_left_to_it(tmp_path):
    reads = _reads(50)
    cases = {
        "length": "@a\n%s\n+\n%s\n" % (reads[0], "I" * (len(reads[0]) - 1)),                 # lengths differ
        "caption": "@a\n%s\n+b\n%s\n" % (reads[0], "I" * len(reads[0])),                      # captions differ
        "blank_in_seq": "@a\n%s %s\n+\n%s\n" % (reads[0][:10], reads[0][10:], "I" * len(reads[0])),
        "form_feed": "@a\n%s\x0c%s\n+\n%sI\n" % (reads[0][:10], reads[0][10:], "I" * len(reads[0])),
        "utf8": b"@a \xc3\xa9\n".decode("latin-1") + "%s\n+\n%s\n" % (reads[0], "I" * len(reads[0])),
        "lone_cr": "@a\n%s\r%s\n+\n%s\n" % (reads[0][:10], reads[0][10:], "I" * len(reads[0])),
    }
    for name, text in cases.items():
        p = str(tmp_path / (name + ".fastq"))
        good = "".join("@g%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads))
        with open(p, "w", newline="", encoding="latin-1") as fh:
            fh.write(good + text)
        # small segments: the good records in front are handled, the loop stops at a batch / segment boundary before the odd one
        got = native.FastqFile.stream_count(p, segment_bytes=1 << 14, batch_size=0)
        assert got[2] <= len(good) and got[0] <= len(reads), (name, got)
        with open(p, "w", newline="", encoding="latin-1") as fh:
            fh.write(text)
        with pytest.raises(native.FastqFile.Unsupported):
            native.FastqFile.stream_count(p)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["fasta60", "fastq80", "fastq_crlf", "fastq_mixed"])
def test_driver_outputs_on_wrapped_files(tmp_path, form, monkeypatch):
    rng = random.Random(9)
    det = scanner.factory(kit="PBC096")
    reads = synth.synth_batch(9000, 21, det.layouts, 1, 0, error_rate=0.07, insert_len=250)
    fasta = form.startswith("fasta")
    p = str(tmp_path / ("r." + ("fasta" if fasta else "fastq")))
    _write(p, form, reads, rng)
    monkeypatch.setenv("QCAT_AMD_SEGMENT_BYTES", str(1 << 20))           # several segments, records across their borders
    out = {}
    for route in ("native", "python"):
        if route == "python":
            monkeypatch.setenv("QCAT_AMD_NO_NATIVE_FASTQ", "1")
        d = str(tmp_path / ("bc_" + route))
        buf = io.StringIO()
        dist = cli.qcat_cli(reads_fq=p, kit="PBC096", mode="epi2me", nobatch=False, out=d, min_qual=None, tsv=False, output=None,
                            threads=1, trim=True, adapter_yaml=None, quiet=True, filter_barcodes=False, middle_adapter=False,
                            min_read_length=50, qcat_config=config.get_default_config(), tsv_stream=buf)
        files = {}
        for f in sorted(os.listdir(d)):
            with open(os.path.join(d, f), "rb") as fh:
                files[f] = fh.read()
        tsv = io.StringIO()
        dist2 = cli.qcat_cli(reads_fq=p, kit="PBC096", mode="epi2me", nobatch=False, out=None, min_qual=None, tsv=True, output=None,
                             threads=1, trim=False, adapter_yaml=None, quiet=True, filter_barcodes=False, middle_adapter=False,
                             min_read_length=0, qcat_config=config.get_default_config(), tsv_stream=tsv)
        out[route] = (dist, files, dist2, tsv.getvalue())
    assert out["native"][0] == out["python"][0] and out["native"][2] == out["python"][2]
    assert out["native"][1].keys() == out["python"][1].keys() and len(out["native"][1]) > 20
    for f in out["native"][1]:
        assert out["native"][1][f] == out["python"][1][f], f
    assert out["native"][3] == out["python"][3] and out["native"][3].count("\n") == len(reads) + 1      # (the header line)


def _run_cli(args, stdin_bytes=None, stdin_path=None, env=None):
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); from qcat_amd import cli; cli.main(sys.argv[1:])" % helpers.ROOT if hasattr(helpers, "ROOT") else None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); from qcat_amd import cli; cli.main(sys.argv[1:])" % root
    e = dict(os.environ)
    e.update(env or {})
    kw = {}
    if stdin_path:
        kw["stdin"] = open(stdin_path, "rb")
    p = subprocess.run([sys.executable, "-c", code] + args, input=stdin_bytes, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600, **kw)
    if stdin_path:
        kw["stdin"].close()
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout, p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["plain", "fastq80", "odd_record_in_the_middle"])
def test_stdin_takes_the_native_loop(tmp_path, form):
    """`cat reads.fastq | qcat -b out --trim` (README.md:104 of the reference): a pipe on the native loop, against the Python loop
    on the same bytes; a record only the Python parser takes in the middle of the stream (blank inside the sequence is an error
    -- here: a record whose caption differs is an error too, so the odd one is a tab inside a title, which is NOT plain for the
    fast splitter but fine for both parsers) hands the rest of the stream back without losing a byte."""
    rng = random.Random(4)
    det = scanner.factory(kit="PBC096")
    reads = synth.synth_batch(12000, 22, det.layouts, 1, 0, error_rate=0.07, insert_len=250)
    p = str(tmp_path / "r.fastq")
    if form == "odd_record_in_the_middle":
        with open(p, "w") as fh:
            for i, r in enumerate(reads):
                q = "I" * len(r)
                if i == 7000:
                    fh.write("@read%d\tx\n%s\x0c%s\n+\n%sI\n" % (i, r[:10], r[10:], q))     # a form feed inside the sequence: Biopython keeps it, the native loop leaves it to the Python parser
                else:
                    fh.write("@read%d ch=%d\n%s\n+\n%s\n" % (i, i % 512, r, q))
    else:
        _write(p, "fastq_mixed" if form == "fastq80" else "fastq_plus_title", reads, rng)
    data = open(p, "rb").read()
    seg = {"QCAT_AMD_SEGMENT_BYTES": str(1 << 20)}
    out = {}
    for route, env in (("native", seg), ("python", dict(seg, QCAT_AMD_NO_NATIVE_FASTQ="1"))):
        d = str(tmp_path / ("bc_" + route))
        _run_cli(["-b", d, "--trim", "-k", "PBC096", "--min-read-length", "50"], stdin_bytes=data, env=env)
        files = {}
        for f in sorted(os.listdir(d)):
            with open(os.path.join(d, f), "rb") as fh:
                files[f] = fh.read()
        tsv, err = _run_cli(["--tsv", "-k", "PBC096", "--min-read-length", "0"], stdin_bytes=data, env=env)
        redirected, _ = _run_cli(["--tsv", "-k", "PBC096", "--min-read-length", "0"], stdin_path=p, env=env)     # `qcat < file`
        assert redirected == tsv
        out[route] = (files, tsv, [l for l in err.decode().splitlines() if "barcode" in l or "reads" in l])
    assert out["native"][0].keys() == out["python"][0].keys() and len(out["native"][0]) > 20
    for f in out["native"][0]:
        assert out["native"][0][f] == out["python"][0][f], f
    assert out["native"][1] == out["python"][1] and out["native"][1].count(b"\n") == len(reads) + 1
    assert out["native"][2] == out["python"][2]
