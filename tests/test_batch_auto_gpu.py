"""Kit auto in batch mode with ONE adapter pass (qcat_scan_batch_auto; SURVEY.md 8f rank 1,
qcat/scanner_base.py:662-678 + :714-733): the records must equal those of the reference's two-pass
formulation -- vote over all auto-detect templates, then detect_barcode of every read with the voted
kit's templates only -- computed here by the two separate native calls and by the oracle."""
import time

import numpy as np
import pytest

import custom_kits
import helpers
import oracle_lib
import synth
from qcat_amd import config, native, scanner

pytestmark = pytest.mark.gpu


def _mixed_batch(det, majority, n, seed):
    """reads of three kits, `majority` most frequent, + degenerate reads"""
    lays = det.layouts
    by_kit = {}
    for i, l in enumerate(lays):
        by_kit.setdefault(l.kit, []).append(i)
    reads = []
    kits = [majority] + [k for k in by_kit if k != majority][:2]
    share = [0.6, 0.25, 0.15]
    for k, frac in zip(kits, share):
        idx = by_kit[k]
        t5 = idx[-1]
        t3 = idx[0] if len(idx) > 1 else -1
        reads += synth.synth_batch(int(n * frac), seed + len(reads), lays, t5, t3, error_rate=0.08)
    reads += ["", "A", "N" * 200, reads[0][:140], reads[1][:301]]
    order = np.random.default_rng(seed).permutation(len(reads))
    return [reads[i] for i in order]


def _two_pass(det, reads, cfg):
    """the reference's formulation with the existing entry points: qcat_detect_kit, then qcat_scan_batch
    on a kit made of the voted kit's templates"""
    kit_name, _ = det.detect_kit(reads, cfg)
    kits = det.get_adapters(kit_name)
    return kit_name, det._run(list(reads), kits, cfg)


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x["barcode"] is y["barcode"] and x["adapter"] is y["adapter"]
        assert float(x["barcode_score"]).hex() == float(y["barcode_score"]).hex()
        assert (x["adapter_end"], x["trim5p"], x["trim3p"], x["exit_status"]) == (y["adapter_end"], y["trim5p"], y["trim3p"], y["exit_status"])


@pytest.mark.parametrize("majority", ["PBC096", "RBK004", "NBD104/NBD114", "RAB204/RAB214"])
def test_one_pass_equals_two_pass_and_oracle(majority):
    det = scanner.factory()                                   # kit auto: the 12 auto-detect templates
    cfg = config.qcatConfig()
    reads = _mixed_batch(det, majority, 3000, 77)
    kit_name, want = _two_pass(det, reads, cfg)
    assert kit_name == majority
    kit = det._native_kit(det.layouts, cfg, native.ENDS_BOTH)
    bases, offsets = native.pack_reads(reads)
    got = det._context().scan_auto(kit, bases, offsets)
    assert got is not None, "the shipped kits must take the one-pass route"
    recs, slot = got
    assert kit.descriptor.kit_names[slot] == majority
    _same(det._records_to_dicts(recs, det.layouts), want)
    _same(det.detect_barcode_batch(reads, [None] * len(reads), cfg), want)
    # oracle: detect_barcode with the voted kit's templates; template indices map back to the full list
    sub = det.get_adapters(kit_name)
    o = oracle_lib.scan(det.descriptor(layouts=sub, qcat_config=cfg), reads, threads=8)
    full_index = np.array([det.layouts.index(l) for l in sub] + [-1])
    o_idx = full_index[o["adapter_idx"]]
    assert np.array_equal(o_idx, recs["adapter_idx"])
    for name in ("barcode_idx", "barcode2_idx", "exit_status", "adapter_end", "trim5p", "trim3p", "raw_score", "score_den"):
        assert np.array_equal(o[name], recs[name]), name


def test_vote_ties_and_truncated_quality_list():
    """equal vote counts -> the kit that voted first wins (dict order + stable sort, :657-660); a shorter
    read_qualities list truncates the results but not the vote (R7)"""
    det = scanner.factory()
    cfg = config.qcatConfig()
    lays = det.layouts
    pbc = [i for i, l in enumerate(lays) if l.kit == "PBC096"]
    rbk = [i for i, l in enumerate(lays) if l.kit == "RBK004"]
    a = synth.synth_batch(40, 5, lays, pbc[-1], pbc[0], error_rate=0.0, no_adapter_fraction=0.0)
    b = synth.synth_batch(40, 6, lays, rbk[0], -1, error_rate=0.0, no_adapter_fraction=0.0)
    for first, second in ((a, b), (b, a)):
        reads = [first[0]] + second + first[1:]
        kit_name, want = _two_pass(det, reads, cfg)
        assert kit_name == first_kit(det, first[0], cfg)
        _same(det.detect_barcode_batch(reads, [None] * len(reads), cfg), want)
        _same(det.detect_barcode_batch(reads, [None] * 7, cfg), want[:7])
    assert det.detect_barcode_batch([], [], cfg) == []


def first_kit(det, read, cfg):
    return det.detect_kit([read], cfg)[0]


def test_golden_batches_take_the_one_pass_route():
    det = scanner.factory()
    g = helpers.golden()
    for fname, entry in g["batch_fastq"].items():
        seqs = [s for _, s in helpers.fastq_records(fname)]
        kit = det._native_kit(det.layouts, config.qcatConfig(), native.ENDS_BOTH)
        recs, slot = det._context().scan_auto(kit, *native.pack_reads(seqs))
        assert kit.descriptor.kit_names[slot] == entry["voted_kit"]
        for rec, w in zip(recs, entry["results"]):
            assert helpers.record_as_golden(rec, det.layouts, "epi2me") == w


def test_custom_kits_on_table_kernels_fall_back_to_two_calls(tmp_path):
    """templates outside the generated kernels share adapter slices: qcat_scan_batch_auto says
    QCAT_ERR_UNSUPPORTED and detect_barcode_batch votes and scans in two calls -- same results."""
    rng = __import__("random").Random(3)
    folder = str(tmp_path)
    for k, (name, seq) in enumerate((("KA", "GGTGCTG" + "N" * 24 + "TTAACCTTTCTGTTGGTGCTGATATTGC"),
                                     ("KB", "CCGTGAC" + "N" * 24 + "AGAGTTTGATCATGGCTCAGGATTACC"))):
        bcs = custom_kits.random_barcodes(rng, 8)
        custom_kits.write_kit(folder, name, name, seq, bcs)
    import yaml, os
    for f in os.listdir(folder):
        d = yaml.safe_load(open(os.path.join(folder, f)))
        d["auto_detect"] = True
        yaml.safe_dump(d, open(os.path.join(folder, f), "w"))
    det = scanner.factory(kit_folder=folder)
    assert len(det.layouts) == 2
    cfg = config.qcatConfig()
    reads = synth.synth_batch(300, 9, det.layouts, 0, -1, error_rate=0.05) + synth.synth_batch(100, 10, det.layouts, 1, -1, error_rate=0.05)
    kit = native.NativeKit(det.descriptor(qcat_config=cfg), jit=False)
    assert det._context().scan_auto(kit, *native.pack_reads(reads)) is None
    kit_name, want = _two_pass(det, reads, cfg)
    assert kit_name == "KA"


def test_one_pass_is_cheaper_than_two_calls():
    det = scanner.factory()
    cfg = config.qcatConfig()
    reads = _mixed_batch(det, "PBC096", 60000, 123)
    kit = det._native_kit(det.layouts, cfg, native.ENDS_BOTH)
    bases, offsets = native.pack_reads(reads)
    det._context().scan_auto(kit, bases, offsets)
    _two_pass(det, reads[:100], cfg)                          # warm both routes (kit upload, buffers)
    sub = det._native_kit(det.get_adapters("PBC096"), cfg, native.ENDS_BOTH)
    one, two = [], []
    for _ in range(4):                                         # (the fastest of four of each: one run is at the mercy of the box)
        t0 = time.perf_counter(); det._context().scan_auto(kit, bases, offsets); t1 = time.perf_counter()
        votes = det._context().detect_kit(kit, bases, offsets)
        det._context().scan(sub, bases, offsets); t2 = time.perf_counter()
        one.append(t1 - t0); two.append(t2 - t1)
    assert votes[0].sum() == len(reads)
    assert min(one) < min(two), (one, two)


def test_reads_handed_over_as_pointers_give_the_same_dicts(monkeypatch):
    """Round 4: detect_barcode_batch passes the str objects' own buffers (qcat_scan_batch_auto_ptrs through
    qcat_amd/_pyglue.so) and builds the dicts in C; without the helper -- or with a read the helper does not take -- the list
    is packed as before.  All routes: identical dicts; the device-side kit choice (k_pick_kit) equals the host's vote."""
    if native._pyglue is None:
        pytest.skip("the optional C helper is not built")
    det = scanner.factory()
    cfg = config.qcatConfig()
    reads = _mixed_batch(det, "PBC096", 4000, 5)
    quals = [None] * len(reads)
    fast = det.detect_barcode_batch(reads, quals, cfg)
    kit_name, want = _two_pass(det, reads, cfg)
    assert kit_name == "PBC096"
    _same(fast, want)
    monkeypatch.setattr(native, "_pyglue", None)                # pure-Python conversions, concatenated upload
    _same(det.detect_barcode_batch(reads, quals, cfg), want)
    monkeypatch.undo()
    mixed = list(reads)
    mixed[5] = mixed[5].encode()                                # bytes are taken as they are
    _same(det.detect_barcode_batch(mixed, quals, cfg), want)
    odd = list(reads)
    odd[7] = odd[7][:50] + "é" + odd[7][51:]               # not ASCII: the whole list goes the old way
    got = det.detect_barcode_batch(odd, quals, cfg)
    _same(got[:7] + got[8:], want[:7] + want[8:])
    # small batches (< 256 reads take the whole-read upload) and a batch in which nobody votes
    _same(det.detect_barcode_batch(reads[:100], quals[:100], cfg), _two_pass(det, reads[:100], cfg)[1])
    empty = det.detect_barcode_batch(["", ""], [None, None], cfg)
    assert all(r["barcode"] is None for r in empty)


def test_calls_of_one_shape_replay_a_captured_graph(monkeypatch):
    """Round 4: the second kit-auto call of a shape captures its device work (two adapter / barcode passes, vote, kit choice:
    ~45 launches on twelve streams) and later calls replay it with one hipGraphLaunch.  Everything that depends on the data is
    decided on the device, so a replay over OTHER reads -- another majority kit -- must still give the two-pass records."""
    det = scanner.factory()
    cfg = config.qcatConfig()
    lib = native.HipLibrary.get().lib
    ctx = det._context()
    batches = [_mixed_batch(det, m, 4000, s) for m, s in (("PBC096", 11), ("RBK004", 12), ("NBD104/NBD114", 13))]
    want = [_two_pass(det, r, cfg) for r in batches]

    def run(reads):
        return det.detect_barcode_batch(reads, [None] * len(reads), cfg), None

    def compacted(reads):
        return sum(min(len(r), 300) for r in reads)

    assert len({(len(r), compacted(r)) for r in batches}) == 1, "the three batches must have one shape"
    before = lib.qcat_ctx_graph_replays(ctx.handle)
    order = [0, 0, 1, 2, 0, 1]                                  # plain, capture, then replays over other reads
    for i in order:
        got, _ = run(batches[i])
        _same(got, want[i][1])
    assert lib.qcat_ctx_graph_replays(ctx.handle) - before >= 3
    # another read count: launched kernel by kernel again (and captured anew on its second call)
    mid = lib.qcat_ctx_graph_replays(ctx.handle)
    short = batches[0][:3000]
    _same(run(short)[0], _two_pass(det, short, cfg)[1])
    assert lib.qcat_ctx_graph_replays(ctx.handle) == mid
    # ... and back: the graph of the old shape is gone (its buffers may have moved), two calls later it replays again
    for i in (1, 1, 2):
        _same(run(batches[i])[0], want[i][1])
    monkeypatch.setenv("QCAT_HIP_NO_GRAPH", "1")
    off = lib.qcat_ctx_graph_replays(ctx.handle)
    _same(run(batches[2])[0], want[2][1])
    assert lib.qcat_ctx_graph_replays(ctx.handle) == off


def test_several_batches_in_one_call_vote_one_by_one():
    """Round 4 (the kit-auto file loop): qcat_scan_batches_auto_ptrs scans consecutive batches in one call, each with the kit
    its own reads voted for -- the records and the per-batch kit slots must equal one qcat_scan_batch_auto_ptrs call per batch."""
    if native._pyglue is None:
        pytest.skip("the optional C helper is not built")
    det = scanner.factory()
    cfg = config.qcatConfig()
    ctx = det._context()
    kit = det._native_kit(det.layouts, cfg, native.ENDS_BOTH)
    reads = []
    for m, s in (("PBC096", 21), ("RBK004", 22), ("NBD104/NBD114", 23), ("RAB204/RAB214", 24), ("PBC096", 25)):
        reads += _mixed_batch(det, m, 1500, s)[:1500]
    reads += _mixed_batch(det, "RBK004", 700, 26)[:611]           # a short last batch
    views = native.read_views(reads)
    recs, slots = ctx.scan_batches_auto_views(kit, views, len(reads), 1500)
    assert len(slots) == 6 and len(set(slots.tolist())) >= 4
    for q in range(6):
        part = reads[q * 1500:(q + 1) * 1500]
        one, slot = ctx.scan_auto_views(kit, native.read_views(part), len(part))
        assert slot == slots[q]
        assert np.array_equal(one, recs[q * 1500:q * 1500 + len(part)]), q
    # one batch only (batch_reads >= n): the plain call
    recs1, slots1 = ctx.scan_batches_auto_views(kit, native.read_views(reads[:1500]), 1500, 4000)
    assert len(slots1) == 1 and slots1[0] == slots[0] and np.array_equal(recs1, recs[:1500])


@pytest.mark.parametrize("kit,n", [(None, 2500), ("NBD103/NBD104", 2500), ("RAB204", 900), (None, 20000), (None, 36000)])
def test_adapter_chains_of_a_small_batch_in_one_launch(kit, n, hip_options):
    """Round 6: the static-letter adapter chains of a small batch -- nine launches for the twelve auto-detect templates -- leave as
    ONE launch whose blockIdx.y picks the chain (csrc/static_generated.inc: k_adapter_multi).  Records, counts, every template's
    raw score and end (the traces) and every barcode row: identical to launches of their own (QCAT_HIP_NO_ADAPTER_MULTI=1) and to
    the oracle; 36 000 reads are beyond the switch (four waves per SIMD over all chains) and take their own launches either way.  The same for the barcode groups
    (k_barcode_multi, QCAT_HIP_NO_BARCODE_MULTI=1)."""
    det = scanner.factory(kit=kit)
    cfg = config.qcatConfig()
    if kit is None:
        reads = _mixed_batch(det, "PBC096", n, 5)
    else:
        reads = synth.synth_batch(n, 9, det.layouts, len(det.layouts) - 1, 0, error_rate=0.08, no_adapter_fraction=0.2)
        reads[5], reads[6], reads[7] = "", "N" * 180, "ACGT" * 50
    desc = det.descriptor(qcat_config=cfg)
    bases, offsets = native.pack_reads(reads)
    o_recs, o_cnt, o_traces, o_rows = oracle_lib.scan(desc, reads, counts=True, trace=True, rows=True, threads=8)
    for variant in ("one launch", "launches of their own", "barcode groups in launches of their own"):
        hip_options(NO_ADAPTER_MULTI=1 if variant == "launches of their own" else None,
                    NO_BARCODE_MULTI=None if variant == "one launch" else 1, NO_TINY=1)
        cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
        recs, traces, rows = native.NativeContext(0).scan(native.NativeKit(desc), bases, offsets, counts=cnt, trace=True, rows=True)
        assert recs.tobytes() == o_recs.tobytes(), variant
        assert np.array_equal(cnt, o_cnt), variant
        for name in native.TRACE_DTYPE.names:
            assert np.array_equal(traces[name], o_traces[name]), (variant, name)
        assert np.array_equal(rows, o_rows), variant
        cnt2 = np.zeros(desc.n_count_buckets, dtype=np.int64)             # (and without the traces: the records-only kernels)
        assert native.NativeContext(0).scan(native.NativeKit(desc), bases, offsets, counts=cnt2).tobytes() == o_recs.tobytes(), variant
        assert np.array_equal(cnt2, o_cnt), variant
