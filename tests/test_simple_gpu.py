"""BarcodeScannerSimple on the GPU (QCAT_MODE_SIMPLE; qcat/scanner_simple.py:41-91):
against the reference's own detect_barcode outputs (golden "simple") through the Python drop-in, and
record-for-record / trace-for-trace against the oracle on larger mixed batches."""
import os

import numpy as np
import pytest

import helpers
import oracle_lib
import synth
from qcat_amd import config, native, scanner

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("i", range(4))
def test_simple_scanner_matches_the_reference(i):
    entry = helpers.golden()["simple"][i]
    det = scanner.factory(mode="simple", kit=entry["list"])
    reads = helpers.simple_reads(entry)
    cfg = config.qcatConfig()
    batch = det.detect_barcode_batch(reads, [None] * len(reads), cfg)
    for k, (read, want) in enumerate(zip(reads, entry["results"])):
        got = batch[k] if k % 5 else det.detect_barcode(read, qcat_config=cfg)      # both entry points
        bc = got["barcode"]
        assert (None if bc is None else bc.name) == want["barcode_name"]
        assert (None if bc is None else bc.id) == want["barcode_id"]
        assert (-1 if bc is None else det.barcodes.index(bc)) == want["barcode_index"]
        assert float(got["barcode_score"]).hex() == want["score_hex"]
        assert got["adapter"] is None
        assert (got["adapter_end"], got["trim5p"], got["trim3p"], got["exit_status"]) == \
               (want["adapter_end"], want["trim5p"], want["trim3p"], want["exit_status"])


@pytest.mark.parametrize("which,kit", [("standard", "PBK004/LWB001"), ("extended", "PBC096")])
def test_simple_vs_oracle(which, kit):
    det = scanner.factory(mode="simple", kit=which)
    lays = scanner.factory(kit=kit).layouts
    reads = synth.synth_batch(5000, 55, lays, 1, 0, error_rate=0.1)
    for i in range(0, 600, 3):
        reads[i] = reads[i][:(i * 7) % 400]
    reads += ["", "A", "N" * 300, reads[7].lower(), "ACGT*-RYKM" * 30]
    for ends in (native.ENDS_BOTH, native.ENDS_5P):
        d = det.descriptor(ends=ends, min_read_length=300, trim=True)
        kit_h = native.NativeKit(d)
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        recs, traces, rows = native.NativeContext(0).scan(kit_h, *native.pack_reads(reads), counts=cnt, trace=True, rows=True)
        o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(d, reads, counts=True, trace=True, rows=True, threads=8)
        assert recs.tobytes() == o_recs.tobytes()
        assert np.array_equal(cnt, o_cnt)
        for name in ("window_len", "best_end", "bc_idx", "bc_raw", "adapter_end"):
            assert np.array_equal(traces[name], o_traces[name]), name
        assert np.array_equal(rows[:, 0, :], o_rows[:, 0, :])
        assert (recs["adapter_idx"] == -1).all()
        assert 0.3 < (recs["barcode_idx"] >= 0).mean() < 0.99
        nb = len(d.slot_ids)
        assert cnt[:nb + 1].sum() + cnt[-1] == len(reads)                    # called + none + skipped
        assert cnt[nb + 1] == 0 and cnt[nb + 2] == len(reads) - cnt[-1]      # kit buckets: always "none"
    # pipelined host path (>= 32 768 reads) and scan() of any length through the drop-in
    many = (reads * 8)[:40000]
    d = det.descriptor()
    got = native.NativeContext(0).scan(native.NativeKit(d), *native.pack_reads(many))
    assert got.tobytes() == oracle_lib.scan(d, many, threads=8).tobytes()
    seqs = [reads[1][:400], reads[2], reads[4][:150], ""]
    got = native.NativeContext(0).scan_sequences(native.NativeKit(det.descriptor(ends=native.ENDS_5P)), *native.pack_reads(seqs))
    assert got.tobytes() == oracle_lib.scan_sequences(det.descriptor(ends=native.ENDS_5P), seqs).tobytes()
    one = det.scan(seqs[0], None, [], [], qcat_config=config.qcatConfig())
    assert one["adapter"] is None and one["adapter_end"] == int(got[0]["adapter_end"])


def test_simple_barcodes_from_a_fasta_file(tmp_path):
    fa = tmp_path / "bc.fa"
    fa.write_text(">first barcode\nAAGAAAGTTGTCGGTGTCTTTGTG\n>second\nTCGATTCCGTTTGTAGTCGTCTGT\n>third\nGAGTCTTGTGTCCCAGTTACCAGG\n")
    det = scanner.factory(mode="simple", kit=str(fa))
    assert [b.name for b in det.barcodes] == ["first barcode", "second", "third"] and [b.id for b in det.barcodes] == [1, 2, 3]
    read = "ACGTAGCTAGCATCGATTAGC" * 3 + "TCGATTCCGTTTGTAGTCGTCTGT" + "GATTACA" * 60
    res = det.detect_barcode(read, qcat_config=config.qcatConfig())
    assert res["barcode"].name == "second" and res["barcode_score"] == 100.0 and res["adapter"] is None
    assert res["trim5p"] == 63 + 24 - 1 and res["trim3p"] == len(read)
    with pytest.raises(TypeError):
        scanner.factory(mode="simple", kit=None)


@pytest.mark.parametrize("i", [0, 1])
def test_simple_barcodes_of_unequal_length_match_the_reference_and_the_oracle(i, tmp_path):
    """A barcode FASTA with barcodes of 16 to 29 letters: every barcode aligned with its own length and normalised by
    it (qcat/scanner_base.py:108-119; round 3 refused such a list).  The drop-in's dicts against the unmodified
    reference's (tests/golden/simple_ragged.json), the device records against the oracle on a larger batch."""
    import json
    with open(os.path.join(helpers.GOLDEN, "simple_ragged.json")) as fh:
        entry = json.load(fh)[i]
    fa = tmp_path / "ragged.fasta"
    fa.write_text(entry["fasta"])
    det = scanner.factory(mode="simple", kit=str(fa))
    reads = helpers.simple_reads(entry)
    cfg = config.qcatConfig()
    for read, want in zip(reads, entry["results"]):
        got = det.detect_barcode(read, qcat_config=cfg)
        assert (got["barcode"].name if got["barcode"] else None) == want["barcode_name"]
        assert float(got["barcode_score"]).hex() == want["score_hex"]
        assert (got["adapter_end"], got["trim5p"], got["trim3p"], got["exit_status"]) == (
            want["adapter_end"], want["trim5p"], want["trim3p"], want["exit_status"])
    many = reads * 20
    d = det.descriptor()
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    got = native.NativeContext(0).scan(native.NativeKit(d), *native.pack_reads(many), counts=cnt)
    want, want_cnt = oracle_lib.scan(d, many, counts=True, threads=8)
    assert got.tobytes() == want.tobytes() and np.array_equal(cnt, want_cnt)
    # scan() of sequences LONGER than max_align_length takes qcat_scan_sequences (k_scan_sequences): every barcode with its own
    # length there too, winners compared by normalised score (that branch aligned over the padded row)
    longs = [r + "ACGGTTCA" * 40 for r in reads[:12] if len(r) > 30] + [reads[0][:151], "GATTACA" * 100]
    assert all(len(q) > cfg.max_align_length for q in longs)
    d5 = det.descriptor(ends=native.ENDS_5P)
    got = native.NativeContext(0).scan_sequences(native.NativeKit(d5), *native.pack_reads(longs))
    assert got.tobytes() == oracle_lib.scan_sequences(d5, longs).tobytes()
    assert (got["barcode_idx"] >= 0).sum() >= 3
    for q, rec in zip(longs[:4], got[:4]):
        one = det.scan(q, None, [], [], qcat_config=cfg)
        assert one["adapter_end"] == int(rec["adapter_end"])
        assert (det.barcodes.index(one["barcode"]) if one["barcode"] else -1) == int(rec["barcode_idx"])

