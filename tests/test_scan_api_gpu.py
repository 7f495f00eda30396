"""The boundary's scan() / scan_middle() on the GPU (qcat/scanner_base.py:466-519):
scan() of sequences of any length and scan_middle() called directly, against the reference's own
outputs (tests/golden/golden_vectors.json "long_scan") and record-for-record against the oracle."""
import numpy as np
import pytest

import helpers
import oracle_lib
import synth
from qcat_amd import config, native, scanner

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("i", range(4))
def test_scan_and_scan_middle_match_the_reference(i):
    entry = helpers.golden()["long_scan"][i]
    det = scanner.factory(mode=entry["mode"], kit=entry["kit"])
    cfg = config.qcatConfig()
    seqs, chim = helpers.long_scan_inputs(entry, det.layouts)
    for s, want in zip(seqs, entry["scan"]):
        res = det.scan(s, None, det.layouts, [], qcat_config=cfg)
        got = {"barcode_id": None if res["barcode"] is None else res["barcode"].id,
               "barcode_name": None if res["barcode"] is None else res["barcode"].name,
               "score_hex": float(res["barcode_score"]).hex(),
               "adapter_kit": None if res["adapter"] is None else res["adapter"].kit,
               "adapter_idx": -1 if res["adapter"] is None else [id(l) for l in det.layouts].index(id(res["adapter"])),
               "adapter_end": res["adapter_end"], "trim5p": res["trim5p"], "trim3p": res["trim3p"],
               "exit_status": res["exit_status"]}
        assert got == want
    got = [det.scan_middle(s, entry["scan_middle_kit"], cfg) for s in chim]
    assert got == entry["scan_middle"]
    assert True in got and False in got


@pytest.mark.parametrize("mode,kit,t5,t3", [("epi2me", "PBC096", 1, 0), ("epi2me", "NBD103/NBD104", 1, 0),
                                            ("epi2me", None, 3, 2), ("dual", None, 1, 0), ("epi2me", "VMK001", 0, -1)])
def test_scan_sequences_vs_oracle(mode, kit, t5, t3):
    det = scanner.factory(mode=mode, kit=kit)
    reads = synth.synth_batch(40, 31337, det.layouts, t5, t3, error_rate=0.1)
    seqs = []
    for i, r in enumerate(reads):
        if i % 4 == 0:
            seqs.append(r[: 100 + 37 * i])                   # every length class, incl. <= 150
        elif i % 4 == 1:
            seqs.append(r[150:-150])                         # a read interior
        elif i % 4 == 2:
            seqs.append((r + reads[i - 1]).lower())          # chimera, lower case
        else:
            seqs.append(r[:300] + "N" * 40 + "RYKM*-" + r[300:])
    seqs += ["", "A", "N" * 500]
    d = det.descriptor(ends=native.ENDS_5P)
    kit_h = native.NativeKit(d)
    bases, offsets = native.pack_reads(seqs)
    seqs.append(reads[0] * 7)                                # a few thousand letters: many blocks of 64 rows
    bases, offsets = native.pack_reads(seqs)
    ctx = native.NativeContext(0)
    got = ctx.scan_sequences(kit_h, bases, offsets)
    lib = native.HipLibrary.get().lib
    assert lib.qcat_ctx_tiny_ends(ctx.handle) == len(seqs)   # one wave per alignment (kernels_tiny.inc, round 5) ...
    want = oracle_lib.scan_sequences(d, seqs)
    assert got.tobytes() == want.tobytes()
    native.set_option("NO_TINY", 1)                          # ... and the general kernel, one lane per sequence
    try:
        general = ctx.scan_sequences(kit_h, bases, offsets)
        assert lib.qcat_ctx_tiny_ends(ctx.handle) == 0
    finally:
        native.set_option("NO_TINY", None)
    assert general.tobytes() == want.tobytes()
    # windows up to max_align_length: the same records as the fast-kernel path (scan() of a 5' window)
    short = [s for s in seqs if len(s) <= 150]
    if short:
        b2, o2 = native.pack_reads(short)
        fast = native.NativeContext(0).scan(kit_h, b2, o2)
        slow = native.NativeContext(0).scan_sequences(kit_h, b2, o2)
        assert fast.tobytes() == slow.tobytes()


def test_scan_middle_with_unknown_kit_raises_like_the_reference():
    det = scanner.factory(kit="PBC096")
    with pytest.raises(IndexError):
        det.scan_middle("ACGT" * 200, "no-such-kit", config.qcatConfig())


@pytest.mark.parametrize("mode,kit", [("epi2me", "PBC096"), ("epi2me", "RBK004"), ("dual", None)])
def test_single_read_calls_of_one_shape_replay_a_captured_graph(mode, kit, monkeypatch):
    """Round 5: detect_barcode on ONE read per call -- the reference's library entry (qcat/test/test_barcode.py:84; the driver's
    --no-batch loop, cli.py:504-509) -- with a named kit: calls of one shape (same kit, one read, the same compacted size)
    replay the first such call's launches as a graph (qcat_scan_batch -> api_graph_run).  Every call's record must be the
    oracle's whatever ran before it: other reads, a read of another size (which neither replays nor disturbs the graph), a
    batch call in between."""
    det = scanner.factory(mode=mode, kit=kit)
    lays = det.layouts
    t5, t3 = (1, 0) if len(lays) > 1 else (0, -1)
    reads = synth.synth_batch(60, 4242, lays, t5, t3, error_rate=0.08)
    reads[7] = reads[7][:220]                                  # (shorter than two windows: another compacted size)
    reads[8] = ""
    reads[9] = reads[9][:40]
    d = det.descriptor(ends=native.ENDS_BOTH)
    want = oracle_lib.scan(d, reads, threads=4)
    kit_h = native.NativeKit(d)
    ctx = native.NativeContext(0)
    lib = native.HipLibrary.get().lib
    before = lib.qcat_ctx_graph_replays(ctx.handle)
    for i, r in enumerate(reads):
        got = ctx.scan(kit_h, *native.pack_reads([r]))
        assert got.tobytes() == want[i:i + 1].tobytes(), i
        if i == 30:                                            # a batch of another shape between the single reads
            assert ctx.scan(kit_h, *native.pack_reads(reads[:20])).tobytes() == want[:20].tobytes()
    assert lib.qcat_ctx_graph_replays(ctx.handle) - before >= 40
    monkeypatch.setenv("QCAT_HIP_NO_GRAPH", "1")
    off = lib.qcat_ctx_graph_replays(ctx.handle)
    for i in (3, 4, 5):
        assert ctx.scan(kit_h, *native.pack_reads([reads[i]])).tobytes() == want[i:i + 1].tobytes()
    assert lib.qcat_ctx_graph_replays(ctx.handle) == off


def test_scan_sequences_in_pieces_of_32767():
    """More sequences than one launch of the one-wave kernels takes ((sequence, set) is a grid dimension): pieces, same records."""
    det = scanner.factory(kit="NBD103/NBD104")
    reads = synth.synth_batch(500, 11, det.layouts, 1, 0, error_rate=0.1)
    distinct = [r[:90 + i % 120] for i, r in enumerate(reads)]
    d = det.descriptor(ends=native.ENDS_5P)
    want = oracle_lib.scan_sequences(d, distinct)
    n = 2 * 32767 + 1234
    seqs = [distinct[i % 500] for i in range(n)]
    bases, offsets = native.pack_reads(seqs)
    ctx = native.NativeContext(0)
    got = ctx.scan_sequences(native.NativeKit(d), bases, offsets)
    assert native.HipLibrary.get().lib.qcat_ctx_tiny_ends(ctx.handle) == n
    assert got.tobytes() == np.concatenate([want] * (n // 500 + 1))[:n].tobytes()
