"""GPU parity tests proper: the HIP path (through the C ABI) against
  (a) the committed golden vectors of the reference's own Python (record + per-end trace +
      every per-barcode raw score), and
  (b) the CPU oracle on seeded synthetic batches at sizes the oracle finishes in seconds,
bit-exact (integer path: no tolerance)."""
import ctypes as C

import numpy as np
import pytest

import helpers
import oracle_lib
import synth
from qcat_amd import config, native, scanner

pytestmark = pytest.mark.gpu

CASES = [c["name"] for c in helpers.golden()["cases"]]
_ctx = {}


def ctx():
    if "c" not in _ctx:
        _ctx["c"] = native.NativeContext(0)
    return _ctx["c"]


def hip_scan(det, reads, ends=native.ENDS_BOTH, trace=True, cfg=None):
    d = det.descriptor(qcat_config=cfg, ends=ends)
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    if trace:
        recs, traces, rows = ctx().scan(kit, bases, offsets, counts=cnt, trace=True, rows=True)
    else:
        recs, traces, rows = ctx().scan(kit, bases, offsets, counts=cnt), None, None
    return d, recs, traces, rows, cnt


def assert_same_as_oracle(d, reads, recs, traces, rows, cnt):
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(d, reads, counts=True, trace=True, rows=True, threads=8)
    assert recs.tobytes() == o_recs.tobytes()
    assert np.array_equal(cnt, o_cnt)
    if traces is not None:
        for name in native.TRACE_DTYPE.names:
            assert np.array_equal(traces[name], o_traces[name]), name
        assert np.array_equal(rows, o_rows)


@pytest.mark.parametrize("name", CASES)
def test_golden_case(name):
    case = [c for c in helpers.golden()["cases"] if c["name"] == name][0]
    det = helpers.make_scanner(case["mode"], case["kit"])
    reads = helpers.case_reads(case, det.layouts)
    d, recs, traces, rows, cnt = hip_scan(det, reads)
    helpers.assert_case_matches(case, recs, traces, rows, det.layouts)
    assert_same_as_oracle(d, reads, recs, traces, rows, cnt)


def test_scan_5p_only_golden():
    g = helpers.golden()["scan5p"]
    det = scanner.factory(kit=g["kit"])
    reads = helpers.case_reads({"gen": g["gen"]}, det.layouts)
    d, recs, traces, rows, cnt = hip_scan(det, reads, ends=native.ENDS_5P)
    for rec, want in zip(recs, g["results"]):
        assert helpers.record_as_golden(rec, det.layouts, "epi2me") == want
    o = oracle_lib.scan(d, reads)
    assert recs.tobytes() == o.tobytes()


@pytest.mark.parametrize("mode,kit,t5,t3,e,n", [
    ("epi2me", "NBD103/NBD104", 1, 0, 0.08, 3000),
    ("epi2me", "PBC096", 1, 0, 0.0, 1500),
    ("epi2me", "PBC096", 1, 0, 0.08, 1500),
    ("epi2me", "PBC096", 1, 0, 0.25, 600),
    ("epi2me", "PBK004/LWB001", 1, 0, 0.12, 1500),
    ("epi2me", "RBK004", 0, -1, 0.1, 800),
    ("epi2me", "VMK001", 0, -1, 0.1, 800),
    ("epi2me", "RAB204/RAB214", 1, 0, 0.1, 600),
    ("epi2me", None, 3, 2, 0.08, 400),
    ("dual", None, 1, 0, 0.08, 800),
    ("dual", None, 1, 0, 0.2, 400),
    ("epi2me", "DUAL", 1, 0, 0.08, 300),
])
def test_synthetic_vs_oracle(mode, kit, t5, t3, e, n):
    det = helpers.make_scanner(mode, kit)
    reads = synth.synth_batch(n, 1234567 + n, det.layouts, t5, t3, error_rate=e)
    d, recs, traces, rows, cnt = hip_scan(det, reads)
    assert_same_as_oracle(d, reads, recs, traces, rows, cnt)
    assert cnt.sum() == 2 * n


def test_ragged_and_degenerate_reads():
    det = scanner.factory(kit="PBC096")
    body = synth.synth_read(3, 5, det.layouts, 1, 0, error_rate=0.05)
    reads = ["", "A", "AC", "N" * 10, "n" * 300, body[:1], body[:38], body[:149], body[:150], body[:151],
             body[:299], body[:300], body, body.lower(), "R" * 200, body[:60] + "N" * 30 + body[90:],
             "*" * 40, body[:75] * 2]
    reads = reads * 5 + [body[:k] for k in range(0, 200, 3)]
    d, recs, traces, rows, cnt = hip_scan(det, reads)
    assert_same_as_oracle(d, reads, recs, traces, rows, cnt)


def test_non_default_config_takes_the_generic_device_path():
    """Affine gaps (open != extend) and a shorter window are valid qcat configurations the
    packed kernels do not cover: they must still run on the GPU and match the oracle."""
    cfg = config.qcatConfig()
    cfg.gap_open = 3
    cfg.gap_extend = 1
    cfg.max_align_length = 120
    cfg.extracted_barcode_extension = 7
    cfg.barcode_context_length = 9
    det = scanner.factory(kit="NBD103/NBD104")
    reads = synth.synth_batch(300, 99, det.layouts, 1, 0, error_rate=0.1)
    d, recs, traces, rows, cnt = hip_scan(det, reads, cfg=cfg)
    assert_same_as_oracle(d, reads, recs, traces, rows, cnt)


def test_scanner_api_returns_reference_shaped_dicts():
    inl = helpers.inline_reads()
    det = scanner.factory(kit="RBK001")
    res = det.detect_barcode(inl["read"])
    assert set(res) == {"barcode", "barcode_score", "adapter", "adapter_end", "trim5p", "trim3p", "exit_status"}
    assert res["barcode"].name == "barcode02" and res["adapter"].kit == "RBK001"
    assert res["barcode_score"] == 100.0 and res["adapter_end"] == 101 and (res["trim5p"], res["trim3p"]) == (101, 993)
    assert det.detect_barcode("")["barcode"] is None
    five = [inl[n] for n in ("read", "read_bc3_exact", "read_bc3", "real_bc03_porechop", "read_nobc")]
    batch = det.detect_barcode_batch(five, [None] * 5)
    want = [b for b in helpers.golden()["batch"] if b["kit"] == "RBK001"][0]["results"]
    for got, w in zip(batch, want):
        assert (got["barcode"].name if got["barcode"] else None) == w["barcode_name"]
        assert float(got["barcode_score"]).hex() == w["score_hex"]
        assert (got["adapter_end"], got["trim5p"], got["trim3p"], got["exit_status"]) == \
               (w["adapter_end"], w["trim5p"], w["trim3p"], w["exit_status"])
    # R7: the default read_qualities=[None] truncates the batch to one read
    assert len(det.detect_barcode_batch(five)) == 1
    dual = scanner.factory(mode="dual")
    r = dual.detect_barcode(inl["real_double_barcode_read"])
    assert r["barcode"].name == "barcode06/95" and r["barcode"].id == "6/95"
    assert r["barcode_score"] == 84.78260869565217


def test_batch_mode_kit_vote_golden():
    """detect_kit vote + detect_barcode_batch under kit auto (CLI default)."""
    inl = helpers.inline_reads()
    five = [inl[n] for n in ("read", "read_bc3_exact", "read_bc3", "real_bc03_porechop", "read_nobc")]
    g = helpers.golden()
    for entry in g["batch"]:
        det = scanner.factory(kit=entry["kit"])
        kit_name, _ = det.detect_kit(five)
        assert kit_name == entry["voted_kit"]
        res = det.detect_barcode_batch(five, [None] * 5)
        lays = det.get_adapters(kit_name)
        for got, w in zip(res, entry["results"]):
            assert (got["barcode"].name if got["barcode"] else None) == w["barcode_name"]
            assert float(got["barcode_score"]).hex() == w["score_hex"]
            assert (got["adapter"].kit if got["adapter"] else None) == w["adapter_kit"]
            assert (got["adapter_end"], got["trim5p"], got["trim3p"], got["exit_status"]) == \
                   (w["adapter_end"], w["trim5p"], w["trim3p"], w["exit_status"])
    det = scanner.factory()
    for fname, entry in g["batch_fastq"].items():
        seqs = [s for _, s in helpers.fastq_records(fname)]
        kit_name, _ = det.detect_kit(seqs)
        assert kit_name == entry["voted_kit"], fname
        res = det.detect_barcode_batch(seqs, [None] * len(seqs))
        for got, w in zip(res, entry["results"]):
            assert (got["barcode"].id if got["barcode"] else None) == w["barcode_id"]
            assert float(got["barcode_score"]).hex() == w["score_hex"]
            assert (got["adapter_end"], got["trim5p"], got["trim3p"], got["exit_status"]) == \
                   (w["adapter_end"], w["trim5p"], w["trim3p"], w["exit_status"])


def test_kit_votes_vs_oracle():
    det = scanner.factory()                       # the 12 auto-detect templates
    reads = []
    for t5, t3, seed in ((3, 2, 11), (5, 4, 12), (1, 0, 13), (9, -1, 14), (10, -1, 15)):
        reads += synth.synth_batch(120, seed, det.layouts, t5, t3, error_rate=0.1)
    reads += ["", "ACGT", "N" * 300]
    d = det.descriptor()
    want_votes, want_pick = oracle_lib.detect_kit_votes(d, reads)
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    votes, first = ctx().detect_kit(kit, bases, offsets)
    assert list(votes) == list(want_votes[:len(det.layouts)])
    for t in range(len(det.layouts)):
        idx = np.nonzero(want_pick == t)[0]
        assert first[t] == (idx[0] if len(idx) else len(reads))


def test_generic_device_path_matches_packed_path():
    """QCAT_HIP_FORCE_GENERIC=1 routes a packed-eligible kit through k_scan_generic: both device
    paths must produce identical records, traces and per-barcode rows."""
    import os
    det = scanner.factory(kit="PBK004/LWB001")
    reads = synth.synth_batch(400, 31337, det.layouts, 1, 0, error_rate=0.1) + ["", "AC", "N" * 77]
    d = det.descriptor()
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    a = ctx().scan(kit, bases, offsets, trace=True, rows=True)
    native.set_option("FORCE_GENERIC", 1)                   # (read when a context is created)
    try:
        gctx = native.NativeContext(0)
    finally:
        native.set_option("FORCE_GENERIC", None)
    b = gctx.scan(kit, bases, offsets, trace=True, rows=True)
    assert a[0].tobytes() == b[0].tobytes()
    for name in native.TRACE_DTYPE.names:
        assert np.array_equal(a[1][name], b[1][name]), name
    assert np.array_equal(a[2], b[2])


@pytest.mark.parametrize("i", range(3))
def test_detect_middle_golden_and_oracle(i):
    """--detect-middle: exit_status 997 etc. against the reference fixtures, and a
    larger chimeric batch against the oracle."""
    entry = helpers.golden()["middle"][i]
    det = scanner.factory(mode=entry["mode"], kit=entry["kit"], scan_middle_adapter=True)
    reads = helpers.middle_reads(entry, det.layouts)
    d = det.descriptor()
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    recs = ctx().scan(kit, bases, offsets, counts=cnt)
    for rec, want in zip(recs, entry["results"]):
        assert helpers.record_as_golden(rec, det.layouts, entry["mode"]) == want
    # dict-level API
    res = det.detect_barcode_batch(reads, [None] * len(reads)) if entry["kit"] else [det.detect_barcode(r) for r in reads]
    assert [r["exit_status"] for r in res] == [w["exit_status"] for w in entry["results"]]
    # bigger batch vs oracle (records + counts)
    g = entry["gen"]
    base = synth.synth_batch(300, 777 + i, det.layouts, g["tpl_5p"], g["tpl_3p"], error_rate=0.08)
    # a read joined to itself keeps one barcode at both ends (no 1002) and has adapters inside
    big = [base[j] + base[j] if j % 2 else base[j] for j in range(150)] + ["", "ACGT" * 80, "N" * 700]
    bases, offsets = native.pack_reads(big)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    recs = ctx().scan(kit, bases, offsets, counts=cnt)
    o_recs, o_cnt = oracle_lib.scan(d, big, counts=True, threads=8)
    assert recs.tobytes() == o_recs.tobytes()
    assert np.array_equal(cnt, o_cnt)
    assert (recs["exit_status"] == 997).sum() > 10


def test_empty_batch_and_api_misuse():
    det = scanner.factory(kit="PBC096")
    assert det.detect_barcode_batch([], []) == []
    d = det.descriptor()
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads([])
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    recs = ctx().scan(kit, bases, offsets, counts=cnt)
    assert len(recs) == 0 and cnt.sum() == 0
    hip = native.HipLibrary.get()
    # non-monotonic offsets are rejected, not read out of bounds
    bad = np.array([0, 10, 5], dtype=np.uint64)
    out = np.zeros(2, dtype=native.RESULT_DTYPE)
    rc = hip.lib.qcat_scan_batch(ctx().handle, kit.handle, np.zeros(16, dtype=np.uint8).ctypes.data, bad.ctypes.data, 2,
                                 out.ctypes.data, None)
    assert rc == -1 and b"non-decreasing" in hip.lib.qcat_last_error()
    assert hip.lib.qcat_scan_batch(None, kit.handle, None, offsets.ctypes.data, 0, out.ctypes.data, None) == -1
    # a 5'-only kit cannot be used for the kit vote
    k5 = native.NativeKit(det.descriptor(ends=native.ENDS_5P))
    with pytest.raises(RuntimeError, match="QCAT_ENDS_BOTH"):
        ctx().detect_kit(k5, *native.pack_reads(["ACGT" * 50]))
    # fetching results of a different batch size than the last scan is an error
    out3 = np.zeros(3, dtype=native.RESULT_DTYPE)
    assert hip.lib.qcat_ctx_fetch_results(ctx().handle, out3.ctypes.data, 3) == -1


def test_host_buffer_scan_uploads_windows_only():
    """qcat_scan_batch compacts every read to head + tail on the host (batches >= 4096 reads): same
    records, traces and counts as the oracle, for both end modes, with reads around the 150 / 300-nt
    boundaries where head and tail overlap."""
    det = scanner.factory(kit="NBD103/NBD104")
    reads = synth.synth_batch(4400, 2024, det.layouts, 1, 0, error_rate=0.08)
    for i, cut in enumerate((0, 1, 149, 150, 151, 299, 300, 301, 302, 449, 450, 451)):
        reads[i] = reads[100 + i][:cut]
    for ends in (native.ENDS_BOTH, native.ENDS_5P):
        d, recs, traces, rows, cnt = hip_scan(det, reads, ends=ends)
        assert_same_as_oracle(d, reads, recs, traces, rows, cnt)


@pytest.mark.parametrize("kit_name,mode,ends", [("NBD103/NBD104", "epi2me", native.ENDS_5P), ("PBC096", "epi2me", native.ENDS_BOTH),
                                                (None, "dual", native.ENDS_BOTH)])
def test_host_buffer_scan_pipeline_equals_one_shot_and_oracle(kit_name, mode, ends, monkeypatch):
    """qcat_scan_batch on >= 32 768 reads runs as a chunked pipeline (host compaction | H2D | kernels,
    host_pipeline.inc): records and counts must equal the one-shot path's and the oracle's, for chunk
    sizes that do and do not divide the batch, with ragged reads at chunk borders, and repeated calls
    on one context must reuse its staging without leaking state."""
    det = scanner.factory(mode=mode, kit=kit_name)
    n = 70001
    reads = synth.synth_batch(n, 4242, det.layouts, 1, 0, error_rate=0.08)
    for i, cut in enumerate((0, 1, 149, 150, 151, 299, 300, 301, 302, 449, 450, 451)):
        for base in (0, 4096 - 6, 8192 - 6, 32768 - 6, n - 12):
            reads[base + i] = reads[100 + i][:cut]
    d = det.descriptor(ends=ends)
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    want, want_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    ctx = native.NativeContext(0)
    monkeypatch.setenv("QCAT_HIP_NO_PIPELINE", "1")
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    one_shot = ctx.scan(kit, bases, offsets, counts=cnt)
    assert one_shot.tobytes() == want.tobytes() and np.array_equal(cnt, want_cnt)
    monkeypatch.delenv("QCAT_HIP_NO_PIPELINE")
    out = np.empty(n, dtype=native.RESULT_DTYPE)
    for chunk in ("4096", "10000", "32768", None, "4096"):
        if chunk:
            monkeypatch.setenv("QCAT_HIP_PIPELINE_CHUNK", chunk)
        else:
            monkeypatch.delenv("QCAT_HIP_PIPELINE_CHUNK")
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        out[:] = 0
        got = ctx.scan(kit, bases, offsets, counts=cnt, out=out)
        assert got is out and got.tobytes() == want.tobytes(), chunk
        assert np.array_equal(cnt, want_cnt), chunk
    # a smaller batch afterwards on the same context (one-shot path) and a resident scan still work
    small = ctx.scan(kit, *native.pack_reads(reads[:5000]))
    assert small.tobytes() == want[:5000].tobytes()


@pytest.mark.parametrize("mode,kit,min_len,trim,middle", [("epi2me", "PBC096", 100, True, False), ("epi2me", "PBC096", "median", True, False),
                                                          ("epi2me", "PBC096", "median", False, False), ("dual", None, "median", True, False),
                                                          ("epi2me", "NBD103/NBD104", "median", True, True)])
def test_skipped_bucket_on_the_device(mode, kit, min_len, trim, middle):
    """the device histogram applies the driver's min-length filter (qcat/cli.py:521-534) exactly like the
    oracle and like a Python restatement of the driver loop -- resident scans, host-buffer scans (one-shot
    and pipelined), and the k_count path of --detect-middle."""
    det = scanner.factory(mode=mode, kit=kit, scan_middle_adapter=middle)
    reads = synth.synth_batch(40000, 909, det.layouts, 1, 0, error_rate=0.08)
    for i in range(0, 4000, 7):
        reads[i] = reads[i][:60 + (i * 13) % 800]
    reads[5], reads[6] = "", "ACGT" * 30
    if min_len == "median":
        min_len = helpers.median_kept_length(det, reads[:2000], trim)
    d = det.descriptor(min_read_length=min_len, trim=trim)
    kit_h = native.NativeKit(d)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    recs = native.NativeContext(0).scan(kit_h, *native.pack_reads(reads), counts=cnt)
    o_recs, o_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    assert recs.tobytes() == o_recs.tobytes() and np.array_equal(cnt, o_cnt)
    assert np.array_equal(cnt, helpers.driver_histogram(d, det.layouts, recs, [len(r) for r in reads], min_len, trim))
    assert 0 < cnt[-1] < len(reads)
    cnt2 = np.zeros(d.n_count_buckets, dtype=np.int64)
    native.NativeContext(0).scan(kit_h, *native.pack_reads(reads[:3000]), counts=cnt2)       # one-shot path
    assert np.array_equal(cnt2, oracle_lib.scan(d, reads[:3000], counts=True, threads=8)[1])


@pytest.mark.parametrize("mode,kit", [("epi2me", "PBC096"), ("epi2me", "NBD103/NBD104"), ("dual", None), ("epi2me", None)])
def test_zero_scores_take_the_sequential_argmax(mode, kit, monkeypatch):
    """The barcode kernels of the shipped kits keep one (max, first index) key per work unit instead of every
    raw score; that is the reference's arg-max (scanner_base.py:125-134) unless the LARGEST raw score is exactly
    0, where its `max_score == 0.0` clause makes the outcome depend on the list order -- those alignments are
    redone sequentially (k_barcode_redo).  Windows that score exactly 0 against every target (letters outside
    the alphabet score 0 with everything), against some targets, and mixtures, must equal the oracle and the
    raw-score path (QCAT_HIP_RAWS=1)."""
    det = scanner.factory(mode=mode, kit=kit)
    base = synth.synth_batch(600, 1717, det.layouts, len(det.layouts) - 1, 0, error_rate=0.08)
    reads = []
    for i, r in enumerate(base):
        k = i % 6
        if k == 0:
            reads.append("R" * (5 + i % 300))                       # every score 0: the LAST barcode wins (R2)
        elif k == 1:
            reads.append(r[:150].replace("A", "R").replace("C", "Y") + r[150:])
        elif k == 2:
            reads.append("N" * (1 + i % 40) + "RY" * (i % 90))
        elif k == 3:
            reads.append(r[:20 + i % 30])                           # tiny regions: scores around 0
        elif k == 4:
            reads.append(("ACGT"[i % 4]) * (3 + i % 200))
        else:
            reads.append(r)
    d = det.descriptor()
    want, want_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    bases, offsets = native.pack_reads(reads)
    monkeypatch.setenv("QCAT_HIP_SUMMARY", "1")                     # key path also for the small sets
    for raws in (None, "1"):
        if raws:
            monkeypatch.setenv("QCAT_HIP_RAWS", raws)
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        got = native.NativeContext(0).scan(native.NativeKit(d), bases, offsets, counts=cnt)
        assert got.tobytes() == want.tobytes(), raws
        assert np.array_equal(cnt, want_cnt)
    # the all-zero windows really exercise the clause: the last barcode of the set is called with score 0
    if mode == "epi2me" and kit:
        _, tr = oracle_lib.scan(d, reads[:1], trace=True)
        nb = len(det.layouts[int(tr[0]["used_tpl"])].barcode_set_1)
        assert int(tr[0]["bc_idx"][0]) == nb - 1 and int(tr[0]["bc_raw"][0]) == 0


@pytest.mark.parametrize("mode,kit,ends", [("epi2me", "PBC096", native.ENDS_BOTH), ("epi2me", "NBD103/NBD104", native.ENDS_5P),
                                           ("dual", None, native.ENDS_BOTH), ("epi2me", None, native.ENDS_BOTH),
                                           ("epi2me", "RBK004", native.ENDS_BOTH)])
def test_bit_sliced_barcode_kernels_equal_the_binary16_kernels_and_the_oracle(mode, kit, ends, monkeypatch):
    """Large batches send the full super-tiles of the two hot region lengths through the bit-sliced kernels
    (kernels_bitslice.inc) and the rest through the packed-binary16 kernels; both must give the oracle's records.
    The batch mixes plain reads (hot), reads with N or letters outside the alphabet (never hot), truncated reads
    (other lengths) and degenerate ones.  The timing ring must show that the bit-sliced kernels really ran."""
    det = scanner.factory(mode=mode, kit=kit)
    t5 = len(det.layouts) - 1 if kit else (3 if mode == "epi2me" else 1)
    t3 = 0 if len(det.layouts) > 1 else -1
    if kit is None and mode == "epi2me":
        t3 = 2
    n = 30000
    reads = synth.synth_batch(n, 9090, det.layouts, t5, t3, error_rate=0.08)
    for i in range(0, n, 11):
        r = reads[i]
        k = (i // 11) % 5
        if k == 0:
            reads[i] = r[:40] + "N" + r[41:70] + "NN" + r[72:]
        elif k == 1:
            reads[i] = r[:60] + "R" + r[61:]
        elif k == 2:
            reads[i] = r[:100 + (i % 400)]
        elif k == 3:
            reads[i] = r[:-30] + "n" + r[-29:]
        else:
            reads[i] = r.lower()
    reads[7], reads[8], reads[9] = "", "N" * 400, "ACGT" * 100
    d = det.descriptor(ends=ends)
    want, want_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    bases, offsets = native.pack_reads(reads)
    got = {}
    monkeypatch.setenv("QCAT_HIP_BITSLICE_MIN", "16384")     # (the path is for batches that fill the chip: force it here)
    for variant in ("static letters", "padded last super-tiles", "no producer wave", "side streams", "letters from memory", "off"):
        if variant == "padded last super-tiles":                 # (round 4: what is left of a hot bin as one more, partly filled super-tile)
            monkeypatch.setenv("QCAT_HIP_BITSLICE_PAD", "128")
        elif variant == "no producer wave":                      # (units with an idle wave: the shared columns as a phase of their own)
            monkeypatch.delenv("QCAT_HIP_BITSLICE_PAD")
            monkeypatch.setenv("QCAT_HIP_BS_NO_SOLO", "1")
            monkeypatch.setenv("QCAT_HIP_LEFTOVER_SIDE", "0")
        elif variant == "side streams":                          # (the arrangement of big batches: one stream per target family)
            monkeypatch.delenv("QCAT_HIP_BS_NO_SOLO")
            monkeypatch.delenv("QCAT_HIP_LEFTOVER_SIDE")
            monkeypatch.setenv("QCAT_HIP_BS_SIDE", "1")
        elif variant == "letters from memory":
            monkeypatch.delenv("QCAT_HIP_BS_SIDE")
            monkeypatch.setenv("QCAT_HIP_NO_BS_STATIC", "1")
        elif variant == "off":
            monkeypatch.setenv("QCAT_HIP_NO_BITSLICE", "1")
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        ctx = native.NativeContext(0)
        lib = native.HipLibrary.get().lib
        native.HipLibrary.get().check(lib.qcat_ctx_set_timing(ctx.handle, 1))
        got[variant] = ctx.scan(native.NativeKit(d), bases, offsets, counts=cnt)
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        k = lib.qcat_ctx_last_timing(ctx.handle, names, ms, 16)
        ran = [names[i].decode() for i in range(k)]
        assert ("k_barcode_bitslice" in ran) == (variant != "off"), (variant, ran)
        bad = np.nonzero(got[variant] != want)[0]
        assert len(bad) == 0, (variant, bad[:10], got[variant][bad[:3]], want[bad[:3]])
        assert np.array_equal(cnt, want_cnt)


@pytest.mark.parametrize("n", [1100, 4000, 9000])
def test_small_batches_of_a_96_barcode_kit_on_padded_super_tiles(n, monkeypatch):
    """Round 4 (measured, not the default: DESIGN 9.4): with QCAT_HIP_BITSLICE_PAD=<jobs> a small batch of a big barcode set
    runs its hot jobs on the bit-sliced kernels with one partly filled super-tile per hot bin.  The records are the oracle's
    with the padding and without it, and with the padding the bit-sliced kernels must really have run."""
    det = scanner.factory(mode="epi2me", kit="PBC096")
    reads = synth.synth_batch(n, 4242 + n, det.layouts, 1, 0, error_rate=0.08)
    for i in range(0, n, 13):
        reads[i] = reads[i][:50] + "N" + reads[i][51:] if i % 2 else reads[i][:120 + i % 300]
    d = det.descriptor(ends=native.ENDS_BOTH)
    want, want_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    bases, offsets = native.pack_reads(reads)
    lib = native.HipLibrary.get().lib
    for pad in ("256", None):
        if pad is not None:
            monkeypatch.setenv("QCAT_HIP_BITSLICE_PAD", pad)
        else:
            monkeypatch.delenv("QCAT_HIP_BITSLICE_PAD")
        ctx = native.NativeContext(0)
        native.HipLibrary.get().check(lib.qcat_ctx_set_timing(ctx.handle, 1))
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        got = ctx.scan(native.NativeKit(d), bases, offsets, counts=cnt)
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        k = lib.qcat_ctx_last_timing(ctx.handle, names, ms, 16)
        ran = [names[i].decode() for i in range(k)]
        assert ("k_barcode_bitslice" in ran) == (pad is not None and 2 * n >= 2048), (pad, ran)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (pad, bad[:10], got[bad[:3]], want[bad[:3]])
        assert np.array_equal(cnt, want_cnt)


def test_timing_ring_and_stream_accessor():
    """qcat_ctx_last_timing averages over the scans since the previous call (no sync between scans);
    qcat_ctx_stream hands out the context's stream for stream-ordered RCCL calls."""
    hip = native.HipLibrary.get()
    lib = hip.lib
    c = native.NativeContext(0)
    assert lib.qcat_ctx_stream(c.handle)
    det = scanner.factory(kit="NBD103/NBD104")
    kit = native.NativeKit(det.descriptor(ends=native.ENDS_5P))
    sp = native.SynthParams(seed=3, n_reads=20000, insert_len=600, lead_min=5, lead_max=40, error_rate=0.08,
                            no_adapter_fraction=0.05, tpl_5p=1, tpl_3p=0)
    b = C.c_void_p()
    hip.check(lib.qcat_batch_synthesize(c.handle, kit.handle, C.byref(sp), C.byref(b)))
    names = (C.c_char_p * 16)()
    ms = (C.c_float * 16)()
    assert lib.qcat_ctx_last_timing(c.handle, names, ms, 16) == 0          # timing is off by default
    hip.check(lib.qcat_ctx_set_timing(c.handle, 1))
    for _ in range(70):                                                  # more scans than the ring holds
        hip.check(lib.qcat_scan_resident(c.handle, kit.handle, b))
    k = lib.qcat_ctx_last_timing(c.handle, names, ms, 16)
    got = [names[i].decode() for i in range(k)]
    assert got[0] == "k_pack_windows" and got[-1] == "k_finalize" and "k_barcode_static" in got
    assert all(0.0 < ms[i] < 50.0 for i in range(k))
    assert lib.qcat_ctx_last_timing(c.handle, names, ms, 16) == 0          # drained
    lib.qcat_batch_destroy(b)


def test_one_kit_shared_by_concurrent_contexts():
    """a qcat_kit is immutable and shareable, a qcat_ctx belongs to one host thread: four threads scan
    different batches through their own contexts with the SAME kit handle (ctypes drops the GIL)."""
    import threading
    det = scanner.factory(kit="PBC096")
    d = det.descriptor()
    kit = native.NativeKit(d)
    batches = [synth.synth_batch(1500, 100 + i, det.layouts, 1, 0, error_rate=0.08) for i in range(4)]
    want = [oracle_lib.scan(d, b, threads=4).tobytes() for b in batches]
    got, errors = [None] * 4, []

    def work(i):
        try:
            c = native.NativeContext(0)
            bases, offsets = native.pack_reads(batches[i])
            for _ in range(3):
                got[i] = c.scan(kit, bases, offsets).tobytes()
        except Exception as exc:                      # pragma: no cover - reported below
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("letters", ["compiled in", "from memory"])
def test_bit_sliced_shapes_of_custom_kits(letters, tmp_path, monkeypatch):
    """The bit-sliced kernels cut a target into shared leading columns (0, 4, 8 or 11 of the longer context) and own
    columns, and walk it backwards when the downstream context is the longer one.  The built-in kits only have 8 and
    11 shared columns: custom kits with short flanks cover the other shapes, on the run-time generated static-letter
    kernels and on the letters-from-memory kernels (both need the kit's generated binary16 chains for the tiles the
    super-tiles leave over, so the kits are compiled first)."""
    import custom_kits
    import random
    rng = random.Random(31)
    folder = str(tmp_path)
    # name: (upstream flank, downstream flank, barcode_context_length) -> shared columns / direction
    shapes = {"P0": ("GGTGCTG", "TTAACCTTTCTGTTGG", 3), "P4F": ("GGTCA", "CAG", 11), "P4R": ("TG", "CAGCAC", 11),
              "P8F": ("CGGTGCTGA", "TTAA", 11), "P8R": ("GCTG", "TTAACCTACT", 11), "P11R": ("GGTGCTG", "TTAACCTTTCTGTTGG", 11)}
    for name, (up, dn, _) in shapes.items():
        blen = 28 if name == "P0" else 24           # (a target needs 32 columns for the packed path: 3 + 28 + 3)
        custom_kits.write_kit(folder, name, name, up + "N" * blen + dn, custom_kits.random_barcodes(rng, 20, length=blen))
    monkeypatch.setenv("QCAT_HIP_BITSLICE_MIN", "2048")
    if letters == "from memory":
        monkeypatch.setenv("QCAT_HIP_NO_BS_STATIC", "1")
    lib = native.HipLibrary.get().lib
    for name in shapes:
        det = scanner.factory(kit=name, kit_folder=folder)
        cfg = config.qcatConfig()
        cfg.barcode_context_length = shapes[name][2]
        d = det.descriptor(qcat_config=cfg, ends=native.ENDS_5P)
        kit = native.NativeKit(d, jit=True)
        info = kit.describe()
        assert info["bitslice_groups"] == 0x10001 and info["n_static_groups"] == 1, (name, info)
        reads = synth.synth_batch(12000, 77, det.layouts, 0, -1, error_rate=0.08)
        reads[5], reads[6] = "", "ACGT" * 50
        want, want_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
        bases, offsets = native.pack_reads(reads)
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        ctx = native.NativeContext(0)
        native.HipLibrary.get().check(lib.qcat_ctx_set_timing(ctx.handle, 1))
        got = ctx.scan(kit, bases, offsets, counts=cnt)
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        ran = [names[i].decode() for i in range(lib.qcat_ctx_last_timing(ctx.handle, names, ms, 16))]
        assert "k_barcode_bitslice" in ran, (name, ran)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (name, bad[:10], got[bad[:3]], want[bad[:3]])
        assert np.array_equal(cnt, want_cnt)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,kit,ends", [("epi2me", "PBC096", native.ENDS_BOTH), ("epi2me", "NBD103/NBD104", native.ENDS_5P),
                                           ("epi2me", "PBK004/LWB001", native.ENDS_BOTH), ("epi2me", None, native.ENDS_BOTH),
                                           ("epi2me", "RBK001", native.ENDS_BOTH), ("dual", None, native.ENDS_BOTH),
                                           ("epi2me", "VMK001", native.ENDS_BOTH)])       # (102 columns: four wide stages, round 4)
def test_bit_sliced_adapter_kernels_equal_the_binary16_kernels_and_the_oracle(mode, kit, ends, monkeypatch):
    """The bit-sliced adapter kernels (kernels_abs.inc) take the full-length windows of plain A/C/G/T of a big batch, the
    binary16 kernels of the same kit the 128-end tiles that hold anything else.  A debug scan of a mixed batch (plain
    reads, reads with N / IUPAC letters / lower case, reads shorter than two windows, degenerate reads) compares the
    per-template raw score AND end_query of every window (trace), every per-barcode row and the records with the
    oracle -- with the path forced, and with it switched off.  Both templates of the dual kit have a plan (the
    98-column one compiles to 51 + 47 columns, QAB_T13); a template without one would simply stay on the binary16 kernels
    beside the ones that have one."""
    det = scanner.factory(mode=mode, kit=kit)
    t5 = len(det.layouts) - 1 if kit else (3 if mode == "epi2me" else 1)
    t3 = 0 if len(det.layouts) > 1 else -1
    if kit is None and mode == "epi2me":
        t3 = 2
    n = 9000
    reads = synth.synth_batch(n, 4242, det.layouts, t5, t3, error_rate=0.08)
    for i in range(0, n, 7):
        r = reads[i]
        k = (i // 7) % 6
        if k == 0:
            reads[i] = r[:40] + "N" + r[41:70] + "NN" + r[72:]
        elif k == 1:
            reads[i] = r[:60] + "R" + r[61:]
        elif k == 2:
            reads[i] = r[:20 + (i % 290)]                        # shorter than two windows, many shorter than one
        elif k == 3:
            reads[i] = r[:-30] + "n" + r[-29:]
        elif k == 4:
            reads[i] = r.lower()
        else:
            reads[i] = ("ACG" * 80)[:150 + i % 50]               # tandem repeat: ties in the end-position rule
    reads[3], reads[4], reads[5] = "", "N" * 400, "A" * 300
    d = det.descriptor(ends=ends)
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(d, reads, counts=True, trace=True, rows=True, threads=16)
    bases, offsets = native.pack_reads(reads)
    lib = native.HipLibrary.get().lib
    has_plan = True                                             # (every built-in selection here has a plan for each of its templates)
    # forced with the two-stage plans everywhere, forced with the four-stage plans where a template has one (the
    # medium-batch form, k_adapter_ms), switched off
    for variant in ("forced", "forced4", "off"):
        if variant.startswith("forced"):
            monkeypatch.setenv("QCAT_HIP_ADAPTER_BITSLICE_MIN", "1")
            monkeypatch.setenv("QCAT_HIP_ABS_STAGES", "4" if variant == "forced4" else "2")
        else:
            monkeypatch.setenv("QCAT_HIP_NO_ADAPTER_BITSLICE", "1")
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        ctx = native.NativeContext(0)
        native.HipLibrary.get().check(lib.qcat_ctx_set_timing(ctx.handle, 1))
        recs, traces, rows = ctx.scan(native.NativeKit(d), bases, offsets, counts=cnt, trace=True, rows=True)
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        ran = [names[i].decode() for i in range(lib.qcat_ctx_last_timing(ctx.handle, names, ms, 16))]
        assert ("k_adapter_bitslice" in ran) == (variant.startswith("forced") and has_plan), (variant, ran)
        for name in native.TRACE_DTYPE.names:
            bad = np.nonzero(np.asarray(traces[name] != o_traces[name]).reshape(len(traces), -1).any(axis=1))[0]
            assert len(bad) == 0, (variant, name, bad[:10], traces[name][bad[:3]], o_traces[name][bad[:3]])
        assert np.array_equal(rows, o_rows)
        bad = np.nonzero(recs != o_recs)[0]
        assert len(bad) == 0, (variant, bad[:10], recs[bad[:3]], o_recs[bad[:3]])
        assert np.array_equal(cnt, o_cnt)


@pytest.mark.parametrize("mode,kit,t5,t3,custom", [("epi2me", "PBC096", 1, 0, False), ("epi2me", "NBD103/NBD104", 1, 0, False),
                                                   ("dual", None, 1, 0, False), ("epi2me", "NBD104/NBD114", 1, 0, False),
                                                   ("epi2me", "PBC096", 1, 0, True)])
def test_bit_sliced_units_padded_at_the_front_for_regions_short_of_nominal(mode, kit, t5, t3, custom, monkeypatch):
    """Round 5: a barcode region clipped by its window -- 1 .. BS_PAD_ROWS (12; round 5: 5) bases short of the nominal length (0.64 % of config 3's jobs, 2 %
    of its step on the binary16 kernels) -- shares the nominal units' row count: its alignment starts a few rows late and is
    held at the boundary state until then (csrc/bs_core.h: bs_hold / bs_keep).  Reads whose adapter starts within a few bases of
    the read's end, so that the region in front of the barcode runs out of the window (both ends; the first bases of some reads
    cut away as well), small batches with the path forced and the class's rest taken as a padded super-tile; static letters, letters from memory, the class off; records
    and counts against the oracle, and the diagnostics say that front-padded units did run."""
    cfg = config.qcatConfig()
    if custom:                                            # another geometry: generated kernels through hipRTC
        cfg.barcode_context_length = 9
        cfg.extracted_barcode_extension = 8
    det = scanner.factory(mode=mode, kit=kit)
    n = 9000
    reads = synth.synth_batch(n, 515, det.layouts, t5, t3, error_rate=0.06, lead_min=0, lead_max=8)
    for i in range(0, n, 7):
        reads[i] = reads[i][(i % 13):len(reads[i]) - (i % 14)]     # windows that start / end inside the adapter's first bases (up to BS_PAD_ROWS and beyond)
    reads[3], reads[4] = "", "ACGT" * 30
    d = det.descriptor(qcat_config=cfg)
    want, want_cnt = oracle_lib.scan(d, reads, counts=True, threads=8)
    bases, offsets = native.pack_reads(reads)
    monkeypatch.setenv("QCAT_HIP_BITSLICE_MIN", "2048")
    monkeypatch.setenv("QCAT_HIP_BITSLICE_PAD", "128")
    lib = native.HipLibrary.get().lib
    for variant in ("static letters", "letters from memory", "short class off"):
        if variant == "letters from memory":
            monkeypatch.setenv("QCAT_HIP_NO_BS_STATIC", "1")
        elif variant == "short class off":
            monkeypatch.delenv("QCAT_HIP_NO_BS_STATIC")
            monkeypatch.setenv("QCAT_HIP_BS_NO_SHORT", "1")
        cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
        ctx = native.NativeContext(0)
        got = ctx.scan(native.NativeKit(d, jit=True), bases, offsets, counts=cnt)
        tiles = (C.c_uint32 * 3)()
        native.HipLibrary.get().check(lib.qcat_ctx_barcode_bitslice_tiles(ctx.handle, tiles))
        assert (tiles[0] > 0) == (variant != "short class off") and tiles[1] + tiles[2] > 0, (variant, list(tiles))
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (variant, bad[:10], got[bad[:3]], want[bad[:3]])
        assert np.array_equal(cnt, want_cnt)
