"""SURVEY.md 8f rank 4, the remainder: the reference's module-level alignment helpers on the device (qcat_sg_align,
include/qcat_hip.h; qcat_amd/scanner_base.py: align_adapter, align_adapter_identity, compute_adapter_identity,
eval_adapter_template, find_best_adapter_template, find_highest_scoring_barcode, extract_barcode_region).
score / end_query / end_ref are pinned like every other alignment (rule R1); `matches` / `length` follow the documented
tie order of tests/golden/sg_independent.py -- parity with parasail itself is unpinned for those two numbers."""
import random

import numpy as np
import pytest

import helpers
import oracle_lib
from qcat_amd import config, native, scanner, scanner_base

pytestmark = pytest.mark.gpu


def _pairs(seed, n):
    rng = random.Random(seed)
    qs, ts = [], []
    for _ in range(n):
        L, M = rng.randrange(1, 400), rng.randrange(1, 110)
        s1 = "".join(rng.choice("ACGT" if rng.random() < 0.93 else "ACGTNRYXacgtnx") for _ in range(L))
        s2 = "".join(rng.choice("ACGTN") for _ in range(M))
        if rng.random() < 0.6 and L > M:                       # the target (with errors) somewhere inside the query
            p = rng.randrange(0, L - M + 1)
            s1 = s1[:p] + "".join(c if (c != "N" and rng.random() > 0.12) else rng.choice("ACGT") for c in s2) + s1[p + M:]
        if rng.random() < 0.05:
            s1 = ("AC" * 200)[:L]                              # repeats: ties
        qs.append(s1)
        ts.append(s2)
    return qs, ts


@pytest.mark.parametrize("gaps", [(2, 2), (1, 1), (5, 2), (3, 1)])
@pytest.mark.parametrize("stats", [False, True, native.STATS_PARASAIL5, native.STATS_ROUND3])
def test_device_alignments_equal_the_oracle(gaps, stats):
    cfg = config.qcatConfig()
    ctx = native.NativeContext(0)
    qs, ts = _pairs(17 + gaps[0], 600)
    for matrix in (cfg.matrix, cfg.matrix_barcode):
        got = native.sg_align(ctx, qs, ts, gaps[0], gaps[1], matrix.table, with_stats=stats)
        for i, (q, t) in enumerate(zip(qs, ts)):
            want = oracle_lib.sg_stats(q, t, gaps[0], gaps[1], matrix.table,
                                       rule=native.STATS_PARASAIL6 if stats in (False, True) else stats)
            have = tuple(int(got[i][k]) for k in ("score", "end_query", "end_ref", "matches", "length"))
            assert have[:3] == want[:3], (i, q, t, have, want)
            assert have[3:] == (want[3:] if stats else (0, 0)), (i, q, t, have, want)


def test_reference_known_answer_through_the_helper():
    """qcat/test/test_barcode.py:291-304: RBK001 against `read_bc3_exact` -> template 0, end_query 101."""
    seq = helpers.inline_reads()["read_bc3_exact"]
    det = scanner.factory(kit="RBK001")
    cfg = config.qcatConfig()
    window = scanner_base.extract_align_sequence(seq, False, cfg.max_align_length)
    idx, end, score = scanner_base.find_best_adapter_template(det.layouts, window, cfg)
    assert idx == 0 and end == 101 and score > 90.0
    assert scanner_base.find_best_adapter_template([], window, cfg) == (-1, -1, -1.0)
    assert scanner_base.find_best_adapter_template(det.layouts, "", cfg) == (-1, -1, -1.0)


def test_helpers_agree_with_the_scanner_and_the_oracle():
    det = scanner.factory(kit="PBC096")
    cfg = config.qcatConfig()
    import synth
    reads = synth.synth_batch(12, 99, det.layouts, 1, 0, error_rate=0.05, no_adapter_fraction=0.0)
    for r in reads:
        window = scanner_base.extract_align_sequence(r, False, cfg.max_align_length)
        idx, end, norm = scanner_base.find_best_adapter_template(det.layouts, window, cfg)
        want = det.scan(window, None, det.layouts, None, cfg)
        assert det.layouts[idx] is want["adapter"] or want["adapter"] is None
        lay = det.layouts[idx]
        # the region path of scan() (scanner_epi2me.py:74-82) redone with the helpers gives scan()'s barcode and score
        region = scanner_base.extract_barcode_region(window, lay, 0, end, cfg) if norm > 90.0 else window[:cfg.max_align_length]
        up, dn = lay.get_upstream_context(cfg.barcode_context_length, 0), lay.get_downstream_context(cfg.barcode_context_length, 0)
        bc, q, s, _e = scanner_base.find_highest_scoring_barcode(region, lay.get_barcode_set(0), cfg, up, dn)
        assert bc is want["barcode"] and s == want["barcode_score"]
        # identity of the adapter alignment: the oracle's statistics under the documented tie order
        e2, ident, raw = scanner_base.eval_adapter_template(lay, window, cfg, identity=True)
        o = oracle_lib.sg_stats(window, lay.get_adapter_sequences(), cfg.gap_open, cfg.gap_extend, cfg.matrix.table)
        bl = lay.get_barcode_length(0) + lay.get_barcode_length(1)
        if o[4] < lay.get_adapter_length() * 0.85:
            assert (e2, ident, raw) == (-1, 0.0, -1)
        else:
            assert (e2, raw) == (o[1], o[0]) and ident == float(o[3]) / float(o[4] - bl)
            assert scanner_base.compute_adapter_identity(lay, window, cfg) == ident
        assert scanner_base.eval_adapter_template(lay, window, cfg, identity=False)[::2] == (o[1], o[0])
    assert scanner_base.align_adapter("", "ACGT", cfg) == (None, 0.0) and scanner_base.align_adapter_identity("ACGT", 4, "", 0, cfg) == (None, 0.0)
    assert scanner_base.find_highest_scoring_barcode("", det.layouts[0].get_barcode_set(0), cfg) == (None, 0, 0.0, -1)


@pytest.mark.parametrize("stats", [True, native.STATS_PARASAIL5, native.STATS_ROUND3])
def test_statistics_on_unique_optimal_paths(stats):
    """tests/golden/sg_unique_vectors.json: alignments with exactly one optimal path (counted by the independent DP,
    tests/golden/make_sg_unique.py).  All five numbers follow from the inputs alone, so the device kernel has to report
    them under every rule of the QCAT_STATS_* switch: `matches` / `length` pinned by uniqueness, not by a tie order."""
    import json
    import os
    with open(os.path.join(helpers.GOLDEN, "sg_unique_vectors.json")) as f:
        cases = json.load(f)["cases"]
    cfg = config.qcatConfig()
    ctx = native.NativeContext(0)
    for (go, ge) in sorted({(c["open"], c["extend"]) for c in cases}):
        for name, matrix in (("adapter", cfg.matrix), ("barcode", cfg.matrix_barcode)):
            sel = [c for c in cases if (c["open"], c["extend"]) == (go, ge) and c["matrix"] == name]
            if not sel:
                continue
            got = native.sg_align(ctx, [c["query"] for c in sel], [c["target"] for c in sel], go, ge, matrix.table, with_stats=stats)
            for i, c in enumerate(sel):
                have = tuple(int(got[i][k]) for k in ("score", "end_query", "end_ref", "matches", "length"))
                assert have == (c["score"], c["end_query"], c["end_ref"], c["matches"], c["length"]), (c, have)
