"""Host pipeline parity: qcat_amd.cli.qcat_cli against the outputs of the
reference driver on the four shipped FASTQ files (tests/golden/cli_golden.json, made by
tests/golden/make_cli_golden.py): TSV text, per-barcode / annotated FASTQ files (sha256), and the
end-of-run histogram lines."""
import hashlib
import io
import json
import logging
import os

import pytest

import helpers
from qcat_amd import cli, config

pytestmark = pytest.mark.gpu

with open(os.path.join(helpers.GOLDEN, "cli_golden.json")) as _fh:
    RUNS = json.load(_fh)["runs"]


class _Capture(logging.Handler):
    def __init__(self):
        logging.Handler.__init__(self)
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def _sha(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


@pytest.mark.parametrize("idx", range(len(RUNS)), ids=["%s:%s" % (r["file"], r["variant"]["tag"]) for r in RUNS])
def test_cli_matches_reference_driver(idx, tmp_path):
    run = RUNS[idx]
    v = run["variant"]
    outdir = str(tmp_path / "bc") if v["dir"] else None
    outfile = str(tmp_path / "out.fastq")
    cap = _Capture()
    root = logging.getLogger()
    old_level = root.level
    root.addHandler(cap)
    root.setLevel(logging.INFO)
    buf = io.StringIO()
    reads_fq = os.path.join(helpers.GOLDEN, "data", run["file"])
    if run["file"].startswith("config1_"):
        # BASELINE config 1: the README's 193-read LWB001 example, regenerated from its seed (tests/synth.py)
        import synth
        from qcat_amd import scanner
        reads_fq = str(tmp_path / run["file"])
        with open(reads_fq, "w") as fh:
            fh.write(synth.config1_fastq(scanner.factory(kit=synth.CONFIG1["kit"]).layouts))
    if run["file"].startswith("flags_"):
        # round 5: the file on which --detect-middle / --filter-barcodes change the outcome, regenerated from its seed
        import synth
        from qcat_amd import scanner
        reads_fq = str(tmp_path / run["file"])
        with open(reads_fq, "w") as fh:
            fh.write(synth.flags_fastq(scanner.factory(kit=synth.FLAGS["kit"]).layouts))
    if run["file"].startswith("fasta_of_"):
        # FASTA input: the plain two-line FASTA of a shipped FASTQ file, derived as tests/golden/make_cli_golden.py derives it
        src = os.path.join(helpers.GOLDEN, "data", run["file"][len("fasta_of_"):].replace(".fasta", ".fastq"))
        with open(src) as fh:
            lines = fh.read().split("\n")
        reads_fq = str(tmp_path / run["file"])
        with open(reads_fq, "w") as fh:
            fh.write("".join(">" + lines[i][1:] + "\n" + lines[i + 1] + "\n" for i in range(0, len(lines) - 3, 4)))
    try:
        cli.qcat_cli(reads_fq=reads_fq, kit=run["kit"], mode=v["mode"],
                     nobatch=v["nobatch"], out=outdir, min_qual=None, tsv=v["tsv"],
                     output=None if v["dir"] else outfile, threads=1, trim=v["trim"], adapter_yaml=None,
                     quiet=False, filter_barcodes=v.get("filter", False), middle_adapter=v.get("middle", False), min_read_length=v["min_len"],
                     qcat_config=config.get_default_config(), tsv_stream=buf)
    finally:
        root.removeHandler(cap)
        root.setLevel(old_level)
    assert buf.getvalue() == run["stdout"]
    assert cap.lines == run["log"]
    files = {}
    if outdir:
        for f in sorted(os.listdir(outdir)):
            files[f] = _sha(os.path.join(outdir, f))
    elif os.path.exists(outfile):
        files["out.fastq"] = _sha(outfile)
    assert files == run["files"]


def test_config1_readme_summary_shape():
    """Config 1 (ii): the summary of the 193-read LWB001 file has the shape of the README's example
    (README.md:119-128 of the reference): one kit, one barcode, the adapter-free reads under `none`."""
    run = [r for r in RUNS if r["file"].startswith("config1_") and r["variant"]["tag"] == "dir-auto-readme"][0]
    log = run["log"]
    assert log[0].startswith("Adapters detected in ") and log[0].endswith(" of 193 reads")
    kits = [l.split()[0] for l in log[1:log.index([x for x in log if x.startswith("Barcodes detected")][0])]]
    assert set(kits) <= {"PBK004/LWB001", "none"}
    bars = [l.split()[0] for l in log[log.index([x for x in log if x.startswith("Barcodes detected")][0]) + 1:] if l.split()]
    assert set(b for b in bars if b.startswith("barcode")) == {"barcode01"}
    assert sorted(run["files"]) == ["barcode01.fastq", "none.fastq"]


def test_native_writers_to_a_file_a_pipe_and_an_append_descriptor(tmp_path):
    """The native writers place their pieces with pwrite when the descriptor has a position and fall back to ordered
    write() otherwise (csrc/fastq_host.inc): the TSV of one file through `> file`, through a pipe and through `>> file`
    (behind a line that is already there) must be the same bytes as the Python loop's (QCAT_AMD_NO_NATIVE_FASTQ=1)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "golden", "data", "nbd103.fastq")
    big = str(tmp_path / "big.fastq")
    with open(src, "rb") as fh:
        one = fh.read()
    with open(big, "wb") as fh:
        for _ in range(40):                                  # (several writer blocks' worth of reads would need 16 k; the paths are the same)
            fh.write(one)
    cmd = [sys.executable, "-m", "qcat_amd.cli", "-f", big, "--tsv", "-k", "NBD103/NBD104"]
    env = dict(os.environ, PYTHONPATH=root)

    def run(extra_env, stdout):
        subprocess.check_call(cmd, cwd=root, env=dict(env, **extra_env), stdout=stdout, stderr=subprocess.DEVNULL)

    with open(tmp_path / "direct.tsv", "wb") as fh:
        run({}, fh)
    with open(tmp_path / "python.tsv", "wb") as fh:
        run({"QCAT_AMD_NO_NATIVE_FASTQ": "1"}, fh)
    p = subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    piped = p.stdout.read()
    assert p.wait() == 0
    with open(tmp_path / "append.tsv", "wb") as fh:
        fh.write(b"a line that was there\n")
    with open(tmp_path / "append.tsv", "ab") as fh:
        run({}, fh)
    want = (tmp_path / "python.tsv").read_bytes()
    assert want.count(b"\n") > 100
    assert (tmp_path / "direct.tsv").read_bytes() == want
    assert piped == want
    assert (tmp_path / "append.tsv").read_bytes() == b"a line that was there\n" + want


def test_kit_auto_file_loop_on_several_contexts(tmp_path, monkeypatch):
    """Round 4: the kit-auto file loop (one vote per batch of 4000 reads, qcat/cli.py:500) hands its batches to up to four
    workers with a context each.  A file of mixed kits -- the vote changes from batch to batch -- must give the same TSV and
    the same records with one worker, with four, and through the Python loop over detect_barcode_batch."""
    import numpy as np
    import synth
    from qcat_amd import native, scanner
    det = scanner.factory()
    lays = det.layouts
    by_kit = {}
    for i, l in enumerate(lays):
        by_kit.setdefault(l.kit, []).append(i)
    reads = []
    for b, kit in enumerate(["PBC096", "RBK004", "NBD104/NBD114", "PBC096", "RAB204/RAB214", "PBK004/LWB001", "PBC096"]):
        idx = by_kit[kit]
        other = by_kit["RBK004" if kit != "RBK004" else "PBC096"]
        part = synth.synth_batch(700, 100 + b, lays, idx[-1], idx[0] if len(idx) > 1 else -1, error_rate=0.08)
        part += synth.synth_batch(300, 200 + b, lays, other[-1], other[0] if len(other) > 1 else -1, error_rate=0.08)
        reads += part                                           # batches of 1000: 70 % of one kit, 30 % of another
    reads += synth.synth_batch(317, 999, lays, by_kit["PBC096"][-1], by_kit["PBC096"][0], error_rate=0.08)   # a short last batch
    fq = str(tmp_path / "mixed.fastq")
    with open(fq, "w") as fh:
        for i, r in enumerate(reads):
            fh.write("@r%d ch=%d\n%s\n+\n%s\n" % (i, i % 512, r, "I" * len(r)))
    cfg = config.qcatConfig()
    kit = det._native_kit(lays, cfg, native.ENDS_BOTH)

    def run(workers):
        monkeypatch.setenv("QCAT_HIP_AUTO_WORKERS", str(workers))
        f = native.FastqFile(fq)
        out = str(tmp_path / ("w%d.tsv" % workers))
        with open(out, "wb") as fh:
            recs, skipped, st = f.demux(det._context(), kit, lays, False, batch_size=1000, kit_auto=True, trim=True,
                                        min_read_length=100, tsv_fd=fh.fileno())
        f.close()
        with open(out, "rb") as fh:
            return recs, skipped, fh.read()

    r1, s1, t1 = run(1)
    r4, s4, t4 = run(4)
    assert t1 == t4 and np.array_equal(r1, r4) and np.array_equal(s1, s4)
    assert t1.count(b"\n") == len(reads)
    # the Python loop of the reference's call shape, batch by batch
    want = []
    for first in range(0, len(reads), 1000):
        chunk = reads[first:first + 1000]
        want += det.detect_barcode_batch(chunk, [None] * len(chunk), cfg)
    got = det._records_to_dicts(r4, lays)
    kits_called = set()
    for g, w in zip(got, want):
        assert g["barcode"] is w["barcode"] and g["adapter"] is w["adapter"] and g["exit_status"] == w["exit_status"]
        assert (g["trim5p"], g["trim3p"]) == (w["trim5p"], w["trim3p"])
        if w["adapter"] is not None:
            kits_called.add(w["adapter"].kit)
    assert len(kits_called) >= 4                              # the vote did change between batches
