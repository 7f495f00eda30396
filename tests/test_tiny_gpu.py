"""The handful-of-reads path (qcat_amd/csrc/kernels_tiny.inc: one wave per alignment, lanes along an anti-diagonal) -- what a
single detect_barcode call (qcat/scanner_base.py:521-604; qcat/test/test_barcode.py:84) runs on.  Every golden case of the
reference's Python and seeded batches against the CPU oracle, with EVERY intermediate (per-template raw score and end, region,
every per-barcode raw score) through qcat_scan_debug; the same inputs on the throughput kernels beside it."""
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


def scan(det, reads, cfg=None, trace=True):
    d = det.descriptor(qcat_config=cfg, ends=native.ENDS_BOTH)
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
    if trace:
        recs, traces, rows = ctx().scan(kit, bases, offsets, counts=cnt, trace=True, rows=True)
    else:
        recs, traces, rows = ctx().scan(kit, bases, offsets, counts=cnt), None, None
    tiny = native.HipLibrary.get().lib.qcat_ctx_tiny_ends(ctx().handle)
    return d, recs, traces, rows, cnt, tiny


def same_as_oracle(d, reads, recs, traces, rows, cnt):
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(d, reads, counts=True, trace=True, rows=True, threads=8)
    assert recs.tobytes() == o_recs.tobytes()
    assert np.array_equal(cnt, o_cnt)
    if traces is not None:
        for name in native.TRACE_DTYPE.names:
            assert np.array_equal(traces[name], o_traces[name]), name
        assert np.array_equal(rows, o_rows)


@pytest.mark.parametrize("name", CASES)
def test_golden_cases_on_the_tiny_kernels(name, monkeypatch):
    """Every golden case (records, per-end traces, every per-barcode raw score of the reference's Python) with the whole
    batch on the one-wave-per-alignment kernels, whatever its size."""
    monkeypatch.setenv("QCAT_HIP_TINY_MAX_ENDS", "4096")
    case = [c for c in helpers.golden()["cases"] if c["name"] == name][0]
    det = helpers.make_scanner(case["mode"], case["kit"])
    reads = helpers.case_reads(case, det.layouts)
    if 2 * len(reads) > 4096:
        pytest.skip("more than 2048 reads")
    d, recs, traces, rows, cnt, tiny = scan(det, reads)
    assert tiny == 2 * len(reads)
    helpers.assert_case_matches(case, recs, traces, rows, det.layouts)
    same_as_oracle(d, reads, recs, traces, rows, cnt)


@pytest.mark.parametrize("mode,kit,t5,t3,e", [
    ("epi2me", "NBD103/NBD104", 1, 0, 0.08),
    ("epi2me", "PBC096", 1, 0, 0.0),
    ("epi2me", "PBC096", 1, 0, 0.12),
    ("epi2me", "PBK004/LWB001", 1, 0, 0.12),
    ("epi2me", "RBK004", 0, -1, 0.1),
    ("epi2me", "VMK001", 0, -1, 0.1),                 # templates of more than 64 letters: two columns per lane
    ("epi2me", "RAB204/RAB214", 1, 0, 0.1),
    ("epi2me", None, 3, 2, 0.08),                     # every template of every kit
    ("dual", None, 1, 0, 0.1),
    ("epi2me", "DUAL", 1, 0, 0.08),
])
@pytest.mark.parametrize("n", [1, 2, 7, 32])
def test_small_batches_against_the_oracle_and_the_throughput_kernels(mode, kit, t5, t3, e, n, monkeypatch):
    det = helpers.make_scanner(mode, kit)
    reads = synth.synth_batch(n, 991 + 13 * n, det.layouts, t5, t3, error_rate=e)
    d, recs, traces, rows, cnt, tiny = scan(det, reads)
    assert tiny == 2 * n                                  # the default: batches of up to 20000 alignments
    same_as_oracle(d, reads, recs, traces, rows, cnt)
    monkeypatch.setenv("QCAT_HIP_NO_TINY", "1")
    d2, recs2, traces2, rows2, cnt2, tiny2 = scan(det, reads)
    assert tiny2 == 0 and recs2.tobytes() == recs.tobytes() and np.array_equal(cnt, cnt2) and np.array_equal(rows, rows2)
    for name in native.TRACE_DTYPE.names:
        assert np.array_equal(traces[name], traces2[name]), name


def test_ragged_and_degenerate_reads_on_the_tiny_kernels(monkeypatch):
    """Empty reads, reads shorter than a template, windows of letters outside the alphabet (raw score 0: rule R2), lower case,
    a read that is one window -- one at a time and all together."""
    monkeypatch.setenv("QCAT_HIP_TINY_MAX_ENDS", "4096")
    det = scanner.factory(kit="PBC096")
    body = synth.synth_read(3, 5, det.layouts, 1, 0, error_rate=0.05)
    reads = ["", "A", "AC", "N" * 10, "n" * 300, body[:1], body[:38], body[:149], body[:150], body[:151], body[:299], body[:300],
             body, body.lower(), "R" * 200, body[:60] + "N" * 30 + body[90:], "*" * 40, body[:75] * 2] + [body[:k] for k in range(0, 200, 7)]
    d, recs, traces, rows, cnt, tiny = scan(det, reads)
    assert tiny == 2 * len(reads)
    same_as_oracle(d, reads, recs, traces, rows, cnt)
    for r in reads[:18]:
        d, recs, traces, rows, cnt, tiny = scan(det, [r])
        assert tiny == 2
        same_as_oracle(d, [r], recs, traces, rows, cnt)


def test_affine_gaps_and_the_boundary_of_the_path(monkeypatch):
    """open != extend is outside the tiny kernels (linear gaps only): the general kernel takes the batch.  A batch one read
    end beyond the limit takes the throughput kernels; --detect-middle runs its interior scan after the tiny kernels."""
    cfg = config.qcatConfig()
    cfg.gap_open = 3
    cfg.gap_extend = 1
    det = scanner.factory(kit="NBD103/NBD104")
    reads = synth.synth_batch(5, 42, det.layouts, 1, 0, error_rate=0.1)
    d, recs, traces, rows, cnt, tiny = scan(det, reads, cfg=cfg)
    assert tiny == 0
    same_as_oracle(d, reads, recs, traces, rows, cnt)
    cfg2 = config.qcatConfig()
    cfg2.max_align_length = 120                           # another window length, linear gaps: on the path
    cfg2.extracted_barcode_extension = 7
    d, recs, traces, rows, cnt, tiny = scan(det, reads, cfg=cfg2)
    assert tiny == 10
    same_as_oracle(d, reads, recs, traces, rows, cnt)
    # the default limit counts alignments (waves): 20000 -- a PBC096 read end is 2 templates + 96 barcodes
    det96 = scanner.factory(kit="PBC096")
    for n, on in ((100, True), (103, False)):
        reads96 = synth.synth_batch(n, 43 + n, det96.layouts, 1, 0, error_rate=0.1)
        d, recs, traces, rows, cnt, tiny = scan(det96, reads96)
        assert tiny == (2 * n if on else 0)
        same_as_oracle(d, reads96, recs, traces, rows, cnt)
    # --detect-middle: a read joined to itself carries its adapter in the interior -> 997; the ends on the tiny kernels first
    det_m = scanner.factory(kit="NBD103/NBD104", scan_middle_adapter=True)
    pair = [reads[0] + reads[0], reads[1]]
    dm = det_m.descriptor(ends=native.ENDS_BOTH, scan_middle=True)
    bases, offsets = native.pack_reads(pair)
    got = ctx().scan(native.NativeKit(dm), bases, offsets)
    assert native.HipLibrary.get().lib.qcat_ctx_tiny_ends(ctx().handle) == 4
    assert got.tobytes() == oracle_lib.scan(dm, pair).tobytes() and int(got["exit_status"][0]) == 997


def test_single_read_calls_through_the_scanner_api():
    """detect_barcode on one read at a time -- the reference's own call shape -- equals the batch call and the oracle."""
    det = scanner.factory(kit="PBC096")
    reads = synth.synth_batch(40, 7, det.layouts, 1, 0, error_rate=0.08)
    batch = det.detect_barcode_batch(reads)
    for r, want in zip(reads, batch):
        got = det.detect_barcode(r)
        assert got == want


def test_custom_kits_and_one_end_only(tmp_path):
    """Custom kits (flanks of other lengths, barcodes of 24 and 28 letters, a template of 92 letters) and a 5'-only kit on the
    one-wave kernels: the kernels hold no letters and no shapes, so nothing is compiled for a kit -- any kit a YAML file
    describes runs on them as it is."""
    import random

    import custom_kits
    rng = random.Random(5)
    folder = str(tmp_path)
    shapes = {"T0": ("GGTGCTG", "TTAACCTTTCTGTTGG", 3, 28), "T1": ("GGTCA", "CAG", 11, 24), "T2": ("TG", "CAGCAC", 11, 24),
              "T3": ("ATCGCCTACCGTGACAAGAAAGTTGTCGGTGTCTTTGTG", "TTAACCTACTTGCCTGTCGCTCTATCTTC", 11, 24)}       # (92 columns: two per lane)
    for name, (up, dn, ctx_len, blen) in shapes.items():
        custom_kits.write_kit(folder, name, name, up + "N" * blen + dn, custom_kits.random_barcodes(rng, 20, length=blen))
    for name, (up, dn, ctx_len, blen) in shapes.items():
        det = scanner.factory(kit=name, kit_folder=folder)
        cfg = config.qcatConfig()
        cfg.barcode_context_length = ctx_len
        for ends in (native.ENDS_5P, native.ENDS_BOTH):
            d = det.descriptor(qcat_config=cfg, ends=ends)
            kit = native.NativeKit(d)
            reads = synth.synth_batch(40, 77, det.layouts, 0, -1, error_rate=0.1) + ["", "ACGT" * 50, "N" * 170]
            bases, offsets = native.pack_reads(reads)
            cnt = np.zeros(d.n_count_buckets, dtype=np.int64)
            recs, traces, rows = ctx().scan(kit, bases, offsets, counts=cnt, trace=True, rows=True)
            n_ends = len(reads) * (1 if ends == native.ENDS_5P else 2)
            assert native.HipLibrary.get().lib.qcat_ctx_tiny_ends(ctx().handle) == n_ends
            same_as_oracle(d, reads, recs, traces, rows, cnt)


def test_detect_middle_on_long_reads(monkeypatch):
    """--detect-middle (scanner_base.py:479-519) on reads whose interior is beyond the packed interior scan (more than 16 384
    letters): the one-wave kernels take them (k_midw_*), the general kernel does when they are switched off -- same records as
    the oracle either way; reads of ordinary length in the same batch stay on the packed interior scan."""
    import random

    from qcat_amd import utils
    rng = random.Random(77)
    det = scanner.factory(kit="NBD103/NBD104", scan_middle_adapter=True)
    base = synth.synth_batch(12, 5, det.layouts, 1, 0, error_rate=0.06)

    def rnd(n):
        return "".join(rng.choice("ACGT") for _ in range(n))
    reads = []
    for i, r in enumerate(base):
        kind = i % 4
        inner = {0: rnd(300), 1: r, 2: utils.revcomp(r), 3: r[:120] + rnd(40)}[kind]     # nothing / the read / its reverse complement / its 5' adapter only
        reads.append(r[:220] + rnd(9000 + 1500 * i) + inner + rnd(9000) + r[-220:])
    reads += base[:6] + [base[0] + base[0]]                       # ordinary reads and an ordinary chimera beside them
    d = det.descriptor(ends=native.ENDS_BOTH, scan_middle=True)
    kit = native.NativeKit(d)
    bases, offsets = native.pack_reads(reads)
    want = oracle_lib.scan(d, reads, threads=8)
    assert 997 in want["exit_status"][:12] and (want["exit_status"][:12] != 997).any()
    lib = native.HipLibrary.get().lib
    monkeypatch.setenv("QCAT_HIP_NO_TINY", "1")                   # (the read ends of this small batch on the throughput kernels either way)
    got = ctx().scan(kit, bases, offsets)
    assert lib.qcat_ctx_middle_wave_reads(ctx().handle) == 0
    assert got.tobytes() == want.tobytes()
    monkeypatch.delenv("QCAT_HIP_NO_TINY")
    got = ctx().scan(kit, bases, offsets)
    called_long = int((oracle_lib.scan(det.descriptor(ends=native.ENDS_BOTH, scan_middle=False), reads[:12])["adapter_idx"] >= 0).sum())
    assert lib.qcat_ctx_middle_wave_reads(ctx().handle) == called_long > 0
    assert got.tobytes() == want.tobytes()
