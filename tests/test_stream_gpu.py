"""The driver's per-file loop in bounded memory (qcat_fastq_demux_stream, csrc/fastq_stream.inc; reference:
qcat/cli.py:235-306, :445-563): the file in segments through read | scan | write must give the bytes the whole-file call
(qcat_fastq_demux) and the Python loop give, whatever the segment size and the reader, with --filter-barcodes
(scanner_base.py:690-712) and --detect-middle (:593-595) on the native path as well, and must hand the file back to the
Python parser at a record that is not a plain one."""
import hashlib
import io
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from qcat_amd import cli, config, native, scanner

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_fastq(path, reads, start=0):
    with open(path, "w") as fh:
        for i, r in enumerate(reads):
            fh.write("@r%d ch=%d\tx=1\n%s\n+\n%s\n" % (start + i, i % 512, r, "I" * len(r)))


def _mixed_reads(n_per, seed0=100):
    """batches of one kit each (70 % / 30 %): the vote changes from batch to batch"""
    det = scanner.factory()
    lays = det.layouts
    by_kit = {}
    for i, l in enumerate(lays):
        by_kit.setdefault(l.kit, []).append(i)
    reads = []
    for b, kit in enumerate(["PBC096", "RBK004", "NBD104/NBD114", "PBC096", "RAB204/RAB214", "PBK004/LWB001"]):
        idx = by_kit[kit]
        other = by_kit["RBK004" if kit != "RBK004" else "PBC096"]
        reads += synth.synth_batch(n_per * 7 // 10, seed0 + b, lays, idx[-1], idx[0] if len(idx) > 1 else -1, error_rate=0.08)
        reads += synth.synth_batch(n_per - n_per * 7 // 10, seed0 + 50 + b, lays, other[-1], other[0] if len(other) > 1 else -1, error_rate=0.08)
    reads += synth.synth_batch(n_per // 3, 999, lays, by_kit["PBC096"][-1], by_kit["PBC096"][0], error_rate=0.08)   # a short last batch
    for i in range(0, len(reads), 17):
        reads[i] = reads[i][:(i * 13) % 500 + 1]                 # short and clipped reads
    return det, reads


def _dir_digest(path):
    return {f: hashlib.sha256(open(os.path.join(path, f), "rb").read()).hexdigest() for f in sorted(os.listdir(path))}


def _hist_of(recs, skipped, lays, dual):
    """what qcat_fastq_demux_stream counts, from the records of the whole-file call"""
    keep = skipped == 0
    w0 = max(1, max(len(l.get_barcode_set(0) or ()) for l in lays))
    w1 = max(1, max(len(l.get_barcode_set(1) or ()) for l in lays)) if dual else 1
    bc = np.zeros((len(lays), w0, w1), dtype=np.int64)
    ad = np.zeros(len(lays), dtype=np.int64)
    none = ad_none = 0
    for r in recs[keep]:
        a, b, b2 = int(r["adapter_idx"]), int(r["barcode_idx"]), int(r["barcode2_idx"])
        if a >= 0:
            ad[a] += 1
        else:
            ad_none += 1
        if a >= 0 and b >= 0 and (not dual or b2 >= 0):
            bc[a, b, b2 if dual else 0] += 1
        else:
            none += 1
    return bc, ad, none, ad_none


@pytest.mark.parametrize("mode,kit_auto", [("epi2me", True), ("epi2me", False), ("dual", False)])
def test_segments_give_the_bytes_of_the_whole_file_call(mode, kit_auto, tmp_path):
    if mode == "dual":
        det = scanner.factory(mode="dual")
        lays = det.layouts
        reads = synth.synth_batch(9000, 31, lays, 1, 0, error_rate=0.08)
    elif kit_auto:
        det, reads = _mixed_reads(1000)
        lays = det.layouts
    else:
        det = scanner.factory(kit="PBC096")
        lays = det.layouts
        reads = synth.synth_batch(40000, 32, lays, 1, 0, error_rate=0.08)      # (>= 32768 reads: the chunked host pipeline)
    fq = str(tmp_path / "in.fastq")
    _write_fastq(fq, reads)
    cfg = config.qcatConfig()
    kit = det._native_kit(lays, cfg, native.ENDS_BOTH)
    dual = mode == "dual"
    common = dict(batch_size=1000, kit_auto=kit_auto, trim=True, min_read_length=100)
    f = native.FastqFile(fq)
    os.makedirs(str(tmp_path / "whole"))
    with open(tmp_path / "whole.tsv", "wb") as fh:
        recs, skipped, _st = f.demux(det._context(), kit, lays, dual, tsv_fd=fh.fileno(), out_dir=str(tmp_path / "whole"), **common)
    f.close()
    want_hist = _hist_of(recs, skipped, lays, dual)
    size = os.path.getsize(fq)
    for tag, seg, reader in (("a", size // 7 + 1, 1), ("b", size // 3, 2), ("c", 0, 0), ("d", 40000, 1)):
        os.makedirs(str(tmp_path / tag))
        with open(tmp_path / (tag + ".tsv"), "wb") as fh:
            bc, ad, none, ad_none, st = native.FastqFile.demux_stream(fq, det._context(), kit, lays, dual, tsv_fd=fh.fileno(),
                                                                       out_dir=str(tmp_path / tag), segment_bytes=seg, reader=reader, **common)
        assert st["incomplete"] == 0 and st["next_offset"] == size and st["n_reads"] == len(reads)
        assert st["n_skipped"] == int(skipped.sum())
        assert seg == 0 or st["segments"] >= 3
        assert (tmp_path / (tag + ".tsv")).read_bytes() == (tmp_path / "whole.tsv").read_bytes()
        assert _dir_digest(str(tmp_path / tag)) == _dir_digest(str(tmp_path / "whole"))
        assert np.array_equal(bc, want_hist[0]) and np.array_equal(ad, want_hist[1]) and (none, ad_none) == want_hist[2:]
    assert want_hist[0].sum() > len(reads) // 3


def _run_cli(fq, tmp_path, tag, native_path, monkeypatch, **kw):
    """qcat_cli with TSV + per-barcode files; returns (tsv text, file digests, return value)"""
    if native_path:
        monkeypatch.delenv("QCAT_AMD_NO_NATIVE_FASTQ", raising=False)
    else:
        monkeypatch.setenv("QCAT_AMD_NO_NATIVE_FASTQ", "1")
    out = str(tmp_path / tag)
    buf = io.StringIO()
    args = dict(reads_fq=fq, kit="auto", mode="epi2me", nobatch=False, out=out, min_qual=None, tsv=True, output=None, threads=1,
                trim=True, adapter_yaml=None, quiet=True, filter_barcodes=False, middle_adapter=False, min_read_length=100,
                qcat_config=config.get_default_config(), tsv_stream=buf)
    args.update(kw)
    ret = cli.qcat_cli(**args)
    return buf.getvalue(), _dir_digest(out), ret


@pytest.mark.parametrize("kit,mode", [("auto", "epi2me"), ("PBC096", "epi2me"), ("auto", "dual")])
def test_filter_barcodes_on_the_native_path_equals_the_python_loop(kit, mode, tmp_path, monkeypatch):
    """--filter-barcodes drops, per batch of the driver's loop, the calls of barcodes seen in at most int(5 % of the most
    frequent key's count) reads -- the "no barcode" key included (scanner_base.py:680-712).  Batches of 500 reads with a skewed
    barcode mix and a short last batch; segments much smaller than the file."""
    monkeypatch.setattr(cli, "BATCH_SIZE", 500)
    monkeypatch.setenv("QCAT_AMD_SEGMENT_BYTES", "300000")
    det = scanner.factory(mode=mode, kit=None if kit == "auto" else kit)
    lays = det.layouts
    t5 = [i for i, l in enumerate(lays) if l.kit in ("PBC096", "DUAL")][-1]
    t3 = [i for i, l in enumerate(lays) if l.kit in ("PBC096", "DUAL")][0]
    reads = []
    for i in range(2300):
        rare = i % 40 == 7
        reads.append(synth.synth_read(i, 77, lays, t5, t3, error_rate=0.06, force_barcode=(i // 40) % 90 + 3 if rare else i % 3))
    fq = str(tmp_path / "skewed.fastq")
    _write_fastq(fq, reads)
    got = _run_cli(fq, tmp_path, "native", True, monkeypatch, kit=kit, mode=mode, filter_barcodes=True)
    want = _run_cli(fq, tmp_path, "python", False, monkeypatch, kit=kit, mode=mode, filter_barcodes=True)
    plain = _run_cli(fq, tmp_path, "plain", True, monkeypatch, kit=kit, mode=mode, filter_barcodes=False)
    assert got == want
    # the filter did drop calls: a dropped call becomes the EMPTY result -- trims 0 / 0 (scanner_base.py:393-407), so with --trim
    # the read is cut to nothing and falls to the minimum-length filter (cli.py:521-530) instead of being counted under "none"
    assert got[0] != plain[0] and got[2][3] > plain[2][3] and len(got[2][0]) < len(plain[2][0])
    assert sum(got[2][0].values()) + got[2][3] == len(reads)


def test_detect_middle_on_the_native_path_equals_the_python_loop(tmp_path, monkeypatch):
    """--detect-middle voids the call of a read whose interior carries a barcoded adapter of the called kit (exit status 997,
    scanner_base.py:479-519, :593-595): chimeric reads (a read joined to itself or to its reverse complement, so that both ends
    carry the same barcode -- two different reads would already leave with the conflict status 1002) between ordinary ones,
    kit auto with a vote per batch and a named kit."""
    monkeypatch.setattr(cli, "BATCH_SIZE", 400)
    monkeypatch.setenv("QCAT_AMD_SEGMENT_BYTES", "500000")
    det = scanner.factory()
    lays = det.layouts
    idx = [i for i, l in enumerate(lays) if l.kit == "NBD104/NBD114"]
    base = synth.synth_batch(1500, 5, lays, idx[-1], idx[0], error_rate=0.05)
    reads = []
    for i, r in enumerate(base):
        if i % 5 == 1:
            reads.append(r + (synth.revcomp_acgt(r) if i % 2 else r))
        else:
            reads.append(r)
    fq = str(tmp_path / "chimeras.fastq")
    _write_fastq(fq, reads)
    for kit in ("auto", "NBD104/NBD114"):
        got = _run_cli(fq, tmp_path, "native_" + kit.replace("/", "_"), True, monkeypatch, kit=kit, middle_adapter=True)
        want = _run_cli(fq, tmp_path, "python_" + kit.replace("/", "_"), False, monkeypatch, kit=kit, middle_adapter=True)
        plain = _run_cli(fq, tmp_path, "plain_" + kit.replace("/", "_"), True, monkeypatch, kit=kit, middle_adapter=False)
        assert got == want
        assert got[2][0].get("none", 0) >= plain[2][0].get("none", 0) + 200          # the chimeras lost their calls


def test_a_record_that_is_not_plain_hands_the_rest_of_the_file_to_the_python_parser(tmp_path, monkeypatch):
    """A record far into the file that only the Python parser takes (a form feed inside its sequence): the native loop ends in
    front of its segment (a batch boundary), the Python parser takes the rest, and the outputs are those of a pure Python run.
    (Round 6: a WRAPPED record no longer ends the native loop -- record 900 is one, and is handled in place.)"""
    monkeypatch.setattr(cli, "BATCH_SIZE", 300)
    monkeypatch.setenv("QCAT_AMD_SEGMENT_BYTES", "400000")
    det = scanner.factory(kit="RBK004")
    reads = synth.synth_batch(2500, 12, det.layouts, 0, -1, error_rate=0.05)
    fq = str(tmp_path / "wrapped.fastq")
    with open(fq, "w") as fh:
        for i, r in enumerate(reads):
            if i == 900:                                       # one record with its sequence and quality on two lines each
                h = len(r) // 2
                fh.write("@r%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:h], r[h:], "I" * h, "I" * (len(r) - h)))
            elif i == 1700:                                    # a title the native loop leaves to the Python parser
                fh.write("@r%d odd\n%s\x0c%s\n+\n%s\n" % (i, r[:10], r[10:], "I" * (len(r) + 1)))     # (a form feed inside the sequence: Biopython keeps it)
            else:
                fh.write("@r%d ch=1\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
    for kw in (dict(kit="RBK004"), dict(kit="auto", filter_barcodes=True)):
        tag = kw["kit"]
        got = _run_cli(fq, tmp_path, "native_" + tag, True, monkeypatch, **kw)
        want = _run_cli(fq, tmp_path, "python_" + tag, False, monkeypatch, **kw)
        assert got == want and got[0].count("\n") - 1 + got[2][3] == len(reads)      # (one row per kept read behind the header)
    # (and the native part did run: the stream stops behind whole batches of 300, in front of the odd record and behind the wrapped one)
    kit = det._native_kit(det.layouts, config.qcatConfig(), native.ENDS_BOTH)
    with open(os.devnull, "wb") as fh:
        st = native.FastqFile.demux_stream(fq, det._context(), kit, det.layouts, False, batch_size=300, kit_auto=True, tsv_fd=fh.fileno(),
                                           segment_bytes=400000)[4]
    assert st["incomplete"] == 1 and 900 < st["n_reads"] <= 1700 and st["n_reads"] % 300 == 0


def test_peak_host_memory_does_not_grow_with_the_file(tmp_path):
    """The reference keeps one batch in memory (qcat/cli.py:235-306); the native loop keeps three segments.  The same command on
    a 5 M-read and on a 20 M-read file (short reads, so that the big one is 6 GB; both are many segments long): the peak
    resident set must not follow the file (the whole-file call of round 3 / 4 kept 57 bytes per read plus the mapping)."""
    det = scanner.factory(kit="RBK004")
    block = synth.synth_batch(20000, 3, det.layouts, 0, -1, error_rate=0.05, insert_len=40)
    text = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(block)).encode()
    peaks = {}
    for name, repeat in (("small", 250), ("big", 1000)):
        fq = str(tmp_path / (name + ".fastq"))
        with open(fq, "wb") as fh:
            for _ in range(repeat):
                fh.write(text)
        code = ("import resource, sys\n"
                "from qcat_amd import cli, config\n"
                "r = cli.qcat_cli(reads_fq=sys.argv[1], kit='RBK004', mode='epi2me', nobatch=False, out=None, min_qual=None, tsv=True, output=None,\n"
                "                 threads=1, trim=True, adapter_yaml=None, quiet=True, filter_barcodes=False, middle_adapter=False,\n"
                "                 min_read_length=0, qcat_config=config.get_default_config(), tsv_stream=open('/dev/null', 'w'))\n"
                "print(r[2], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)\n")
        out = subprocess.run([sys.executable, "-c", code, fq], cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        n, peak_kb = out.stdout.decode().split()[-2:]
        assert int(n) == 20000 * repeat
        peaks[name] = int(peak_kb) / 1024.0
        os.remove(fq)
    # 15 M reads more: 57 bytes per read of index and records alone would be + 0.85 GB, the mapping + 5 GB
    assert peaks["big"] < peaks["small"] + 200, peaks
