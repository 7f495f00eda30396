"""Oracle vs the outputs of the reference's own Python (tests/golden/golden_vectors.json,
made by tests/golden/make_golden.py).  Pins everything above the parasail call.  CPU only."""
import json
import os

import numpy as np
import pytest

import helpers
import oracle_lib
import synth
from qcat_amd import config, native, scanner

CASES = [c["name"] for c in helpers.golden()["cases"]]


@pytest.mark.parametrize("name", CASES)
def test_case(name):
    case = [c for c in helpers.golden()["cases"] if c["name"] == name][0]
    det = helpers.make_scanner(case["mode"], case["kit"])
    assert det.min_quality == case["min_quality"]
    reads = helpers.case_reads(case, det.layouts)
    recs, traces, rows = oracle_lib.scan(det.descriptor(), reads, trace=True, rows=True)
    helpers.assert_case_matches(case, recs, traces, rows, det.layouts)


def test_region_table():
    """extract_barcode_region incl. Python negative-index wrap (R4) through the oracle trace:
    force `stop` by scanning windows where the trace reports best_end, then compare lengths via
    the closed form stored in the fixture."""
    g = helpers.golden()
    cfg = config.qcatConfig()
    for row in g["region_table"]:
        lays = scanner.factory(mode="epi2me", kit=row["kit"]).layouts
        tpl = lays[row["tpl"]]
        L, s = row["L"], row["set"]
        ext = cfg.extracted_barcode_extension
        for stop, want in zip(range(-1, L), row["lens"]):
            end = stop - (tpl.get_adapter_length() - tpl.get_barcode_end(s)) + 1
            start = end - tpl.get_barcode_length(s)
            start -= min(ext, start)
            end += min(ext, L - end)
            assert len(("x" * L)[start:end + 1]) == want


def test_scan_5p_only():
    g = helpers.golden()["scan5p"]
    det = scanner.factory(kit=g["kit"])
    reads = helpers.case_reads({"gen": g["gen"]}, det.layouts)
    recs = oracle_lib.scan(det.descriptor(ends=native.ENDS_5P), reads)
    for rec, want in zip(recs, g["results"]):
        assert helpers.record_as_golden(rec, det.layouts, "epi2me") == want


def test_batch_fixed_kit_matches_per_read():
    """detect_barcode_batch with a fixed kit == detect_barcode per read (batch fixture)."""
    r = helpers.inline_reads()
    five = [r[n] for n in ("read", "read_bc3_exact", "read_bc3", "real_bc03_porechop", "read_nobc")]
    entry = [b for b in helpers.golden()["batch"] if b["kit"] == "RBK001"][0]
    det = scanner.factory(kit="RBK001")
    recs = oracle_lib.scan(det.descriptor(), five)
    for rec, want in zip(recs, entry["results"]):
        assert helpers.record_as_golden(rec, det.layouts, "epi2me") == want


@pytest.mark.parametrize("i", range(3))
def test_detect_middle(i):
    """--detect-middle (scan_middle, qcat/scanner_base.py:479-519): exit_status 997 and the rest of
    the dict against the reference."""
    entry = helpers.golden()["middle"][i]
    det = scanner.factory(mode=entry["mode"], kit=entry["kit"], scan_middle_adapter=True)
    reads = helpers.middle_reads(entry, det.layouts)
    recs = oracle_lib.scan(det.descriptor(), reads)
    assert 997 in [int(r["exit_status"]) for r in recs]
    for rec, want in zip(recs, entry["results"]):
        assert helpers.record_as_golden(rec, det.layouts, entry["mode"]) == want


def test_dp_against_independent_vectors():
    """The oracle's C DP (qo_sg) against 12 000 (score, end_query, end_ref) triples computed by the
    independent scalar Python DP (tests/golden/sg_independent.py via make_sg_vectors.py): adapter
    and barcode alignments as the reference runs them, adapter-free windows (end-position ties of
    rule R1), affine gaps with open != extend, degenerate and repetitive inputs (tests/sg_cases.py)."""
    import json
    import os

    import sg_cases
    with open(os.path.join(helpers.GOLDEN, "sg_vectors.json")) as fh:
        fx = json.load(fh)
    assert fx["n"] >= 10000 and len(fx["results"]) == fx["n"]
    st = fx["stats"]
    assert st["ends_in_last_column"] > 1000 and st["ends_in_last_row"] > 1000 and st["ends_in_corner"] > 50
    bad = []
    for i, want in enumerate(fx["results"]):
        s1, s2, go, ge, table = sg_cases.case(fx["seed"], i)
        got = oracle_lib.sg(s1, s2, go, ge, table)
        if list(got) != want:
            bad.append((i, got, want))
    assert not bad, bad[:5]


def test_fixture_dp_values_do_not_come_from_the_oracle():
    g = helpers.golden()
    assert "independent" in g["dp"] and "sg_independent" in g["dp"]
    src = open(os.path.join(helpers.GOLDEN, "make_golden.py")).read()
    assert "import oracle_lib" not in src and "qo_sg(" not in src
    src = open(os.path.join(helpers.GOLDEN, "sg_independent.py")).read()
    assert "import oracle_lib" not in src and "ctypes" not in src


@pytest.mark.parametrize("i", range(4))
def test_scan_of_long_sequences(i):
    """scan() on sequences longer than max_align_length (scanner_base.py:466-477): the oracle's
    qo_scan_sequences against the reference's own scan() outputs."""
    entry = helpers.golden()["long_scan"][i]
    det = scanner.factory(mode=entry["mode"], kit=entry["kit"])
    seqs, _ = helpers.long_scan_inputs(entry, det.layouts)
    recs = oracle_lib.scan_sequences(det.descriptor(ends=native.ENDS_5P), seqs)
    for rec, want in zip(recs, entry["scan"]):
        assert helpers.record_as_golden(rec, det.layouts, entry["mode"]) == want


@pytest.mark.parametrize("mode,kit,min_len,trim", [("epi2me", "PBC096", 100, True), ("epi2me", "PBC096", "median", True),
                                                   ("epi2me", "PBC096", "median", False), ("dual", None, "median", True),
                                                   ("epi2me", "NBD103/NBD104", 0, True)])
def test_skipped_bucket_follows_the_driver(mode, kit, min_len, trim):
    """[skipped] bucket of the count vector = the reference driver's min-length filter applied after
    trimming (qcat/cli.py:521-534): oracle counts vs a Python restatement of that loop."""
    det = scanner.factory(mode=mode, kit=kit)
    reads = synth.synth_batch(300, 808, det.layouts, 1, 0, error_rate=0.08)
    reads += [r[:k] for r, k in zip(reads[:40], range(60, 860, 20))] + ["", "ACGT" * 30]
    if min_len == "median":                                   # a threshold that splits the batch
        min_len = helpers.median_kept_length(det, reads, trim)
    d = det.descriptor(min_read_length=min_len, trim=trim)
    recs, cnt = oracle_lib.scan(d, reads, counts=True)
    want = helpers.driver_histogram(d, det.layouts, recs, [len(r) for r in reads], min_len, trim)
    assert np.array_equal(cnt, want)
    assert cnt.sum() - cnt[-1] == 2 * (len(reads) - cnt[-1])        # every kept read: one barcode + one kit bucket
    if min_len >= 300:
        assert 0 < cnt[-1] < len(reads)
    if min_len == 0:
        assert cnt[-1] == 0


@pytest.mark.parametrize("i", range(4))
def test_simple_mode(i):
    """BarcodeScannerSimple (qcat/scanner_simple.py): oracle vs the reference's detect_barcode."""
    entry = helpers.golden()["simple"][i]
    det = scanner.factory(mode="simple", kit=entry["list"])
    assert len(det.barcodes) == entry["n_barcodes"] and det.min_quality == entry["min_quality"] == 60
    reads = helpers.simple_reads(entry)
    recs = oracle_lib.scan(det.descriptor(), reads)
    assert [helpers.simple_record_as_golden(r, det.barcodes) for r in recs] == entry["results"]
    assert sum(1 for r in entry["results"] if r["barcode_name"]) > len(reads) // 2


def test_dp_statistics_against_the_independent_dp():
    """qo_sg_stats_rule (matches / alignment length along one optimal path, oracle/qcat_oracle.c) against the independent
    scalar DP's sg_stats on seeded pairs, under every rule of the shared switch (QCAT_STATS_*): parasail's recalled order
    with matches over mapped codes (six- and five-letter alphabets) and round 3's order; same score / end position as
    qo_sg under all of them.  Parity with parasail's own choice of path is unpinned -- nothing on a scanner path consumes
    the two numbers."""
    import random
    import sys
    sys.path.insert(0, os.path.join(helpers.GOLDEN))
    import sg_independent as si
    from qcat_amd import config as qconfig, native
    cfg = qconfig.qcatConfig()
    rng = random.Random(20260929)
    differ = 0
    for matrix, alphabet, rules in ((cfg.matrix, "ATGCNX", (native.STATS_PARASAIL6, native.STATS_ROUND3)),
                                    (cfg.matrix_barcode, "ATGCN", (native.STATS_PARASAIL5, native.STATS_ROUND3))):
        tab = matrix.table
        alpha = "ATGCNX"

        def score(a, b, tab=tab):
            ia, ib = alpha.find(a.upper()), alpha.find(b.upper())
            return int(tab[ib if ib >= 0 else 6][ia if ia >= 0 else 6])
        for _ in range(150):
            L, M = rng.randrange(1, 90), rng.randrange(1, 50)
            s1 = "".join(rng.choice("ACGT" if rng.random() < 0.9 else "ACGTNRYXacgtx") for _ in range(L))
            s2 = "".join(rng.choice("ACGTN" if rng.random() < 0.95 else "RYX") for _ in range(M))
            if rng.random() < 0.5 and L > M:
                p = rng.randrange(0, L - M + 1)
                s1 = s1[:p] + "".join(c if (c != "N" and rng.random() > 0.1) else rng.choice("ACGT") for c in s2) + s1[p + M:]
            go, ge = rng.choice([(2, 2), (1, 1), (3, 1), (5, 2)])
            got = {}
            for rule in rules:
                name = "round3" if rule == native.STATS_ROUND3 else "parasail"
                got[rule] = oracle_lib.sg_stats(s1, s2, go, ge, tab, rule=rule)
                assert tuple(si.sg_stats(s1, s2, go, ge, score, alphabet=alphabet, rule=name)) == got[rule], (s1, s2, go, ge, rule)
                assert got[rule][:3] == oracle_lib.sg(s1, s2, go, ge, tab)
            differ += got[rules[0]] != got[rules[1]]
    assert differ > 0          # the rules are not the same function: some pair walks another path or counts '*' matches


def test_mapped_code_matches_known_cases():
    """What "a match = equal mapped codes" means, on hand-made pairs: two different letters outside the alphabet both map
    to '*' and count under the parasail rules, not under round 3's; X is a letter of the adapter alphabet and '*' under
    the barcode alphabet."""
    from qcat_amd import config as qconfig, native
    cfg = qconfig.qcatConfig()
    tab = cfg.matrix.table
    # ACGT R ACGT against ACGT Y ACGT: every letter on the diagonal (R/Y score 0, a gap would cost)
    a6 = oracle_lib.sg_stats("ACGTRACGT", "ACGTYACGT", 2, 2, tab, rule=native.STATS_PARASAIL6)
    r3 = oracle_lib.sg_stats("ACGTRACGT", "ACGTYACGT", 2, 2, tab, rule=native.STATS_ROUND3)
    assert a6[:3] == r3[:3] and a6[4] == r3[4] == 9 and a6[3] == 9 and r3[3] == 8
    # X against R: different codes over ATGCNX, both '*' over ATGCN
    bt = cfg.matrix_barcode.table
    x6 = oracle_lib.sg_stats("ACGTXACGT", "ACGTRACGT", 1, 1, bt, rule=native.STATS_PARASAIL6)
    x5 = oracle_lib.sg_stats("ACGTXACGT", "ACGTRACGT", 1, 1, bt, rule=native.STATS_PARASAIL5)
    assert x6[:3] == x5[:3] and x6[3] == 8 and x5[3] == 9


def _ragged_entries():
    with open(os.path.join(helpers.GOLDEN, "simple_ragged.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("i", [0, 1])
def test_simple_mode_barcodes_of_unequal_length_against_the_reference(i, tmp_path):
    """A barcode FASTA whose barcodes have different lengths (simple mode): the reference aligns every barcode with its
    own length and normalises by it (qcat/scanner_base.py:108-119).  Fixture = the unmodified reference's
    detect_barcode outputs (tests/golden/make_simple_ragged_golden.py); here the oracle through the descriptor's
    per-barcode lengths (qcat_barcode_set_desc.lengths, ABI 4)."""
    entry = _ragged_entries()[i]
    fa = tmp_path / "ragged.fasta"
    fa.write_text(entry["fasta"])
    det = scanner.factory(mode="simple", kit=str(fa))
    assert [len(b.sequence) for b in det.barcodes] == entry["lengths"] and len(set(entry["lengths"])) > 1
    reads = helpers.simple_reads(entry)
    recs = oracle_lib.scan(det.descriptor(), reads)
    got = [helpers.simple_record_as_golden(r, det.barcodes) for r in recs]
    assert got == entry["results"]
    dens = set(int(r["score_den"]) for r in recs if r["barcode_idx"] >= 0)
    assert len(dens) > 2                                   # winners of several lengths: the denominators travel per barcode


def test_unequal_barcode_lengths_outside_simple_mode_are_a_value_error():
    """a template's placeholder has one length (layout.py:55-61): epi2me / dual kits refuse such a set with ValueError,
    which the driver logs (cli.main) instead of a traceback"""
    from qcat_amd import config as qconfig, native
    det = scanner.factory(mode="epi2me", kit="NBD103/NBD104")
    lay = det.layouts[0]
    import copy
    lay2 = copy.copy(lay)
    bs = list(lay.get_barcode_set(0))
    from qcat_amd.adapters import Barcode
    bs[1] = Barcode(bs[1].name, bs[1].id, bs[1].sequence[:20], bs[1].fwd_strand)
    lay2.barcode_set_1 = bs
    with pytest.raises(ValueError):
        native.KitDescriptor([lay2], qconfig.qcatConfig(), mode="epi2me")


def _unique_cases():
    import json
    with open(os.path.join(helpers.GOLDEN, "sg_unique_vectors.json")) as f:
        return json.load(f)["cases"]


def test_statistics_on_unique_optimal_paths_need_no_tie_rule():
    """tests/golden/sg_unique_vectors.json (make_sg_unique.py): 400 alignments with exactly ONE optimal path.  There
    (matches, length) are facts of the inputs, so the oracle has to give them under EVERY rule of the shared switch --
    these cases pin the two numbers to mathematics; only alignments with tied paths still rest on the recalled order."""
    from qcat_amd import config as qconfig, native
    cfg = qconfig.qcatConfig()
    cases = _unique_cases()
    assert len(cases) == 400 and sum(1 for c in cases if c["length"] != c["matches"]) >= 320
    for c in cases:
        tab = (cfg.matrix if c["matrix"] == "adapter" else cfg.matrix_barcode).table
        want = (c["score"], c["end_query"], c["end_ref"], c["matches"], c["length"])
        for rule in (native.STATS_PARASAIL6, native.STATS_PARASAIL5, native.STATS_ROUND3):
            assert oracle_lib.sg_stats(c["query"], c["target"], c["open"], c["extend"], tab, rule=rule) == want, (c, rule)


def test_path_count_dp_sees_ties():
    """sg_independent.sg_unique_path on hand-made pairs: a deletion inside a homopolymer can sit in several places (not
    unique), a clean embedded copy is unique."""
    import sys
    sys.path.insert(0, os.path.join(helpers.GOLDEN))
    import sg_independent as si
    from qcat_amd import config as qconfig
    score = si.scorer_from_table7(qconfig.qcatConfig().matrix_barcode.table)
    n, st = si.sg_unique_path("GGACGTTCAGG", "ACGTTCA", 1, 1, score)
    assert n == 1 and st == (7, 8, 6, 7, 7)
    n, _ = si.sg_unique_path("GGACAAATCAGG", "ACAAAATCA", 1, 1, score)       # one A short in a run of four: four placements
    assert n == 2
    n, _ = si.sg_unique_path("T", "GTCT", 1, 1, score)                         # best score shared by two border cells
    assert n == 2
    n, st = si.sg_unique_path("ACGT", "TTTT", 1, 1, score)                     # only the first target letter can pair with the last T
    assert n == 1 and st == (1, 3, 0, 1, 1)


# ---- rule R1 under the reference's OTHER alignment routine (round 6) ------------------------------------------------
R1_CASES = [c["name"] for c in helpers.golden_r1_scalar()["cases"]]


@pytest.mark.parametrize("name", R1_CASES)
def test_case_r1_scalar(name):
    """the oracle with QCAT_R1_SCALAR against the unmodified reference run with plain `parasail.sg` bound
    (qcat/scanner_base.py:20-26; tests/golden/make_golden.py --r1-scalar)"""
    case = [c for c in helpers.golden_r1_scalar()["cases"] if c["name"] == name][0]
    det = helpers.make_scanner(case["mode"], case["kit"])
    reads = helpers.case_reads(case, det.layouts)
    with helpers.r1_rule("scalar"):
        desc = det.descriptor()
    assert desc.desc.r1_rule == native.R1_SCALAR
    recs, traces, rows = oracle_lib.scan(desc, reads, trace=True, rows=True)
    helpers.assert_case_matches(case, recs, traces, rows, det.layouts)


def test_r1_rules_part_on_adapter_free_reads():
    """the two rules are not the same function: on the adapter-free fixture reads some end positions differ (and no score does)"""
    case = [c for c in helpers.golden_r1_scalar()["cases"] if c["name"] == "synth:NBD104:e0.08:bare"][0]
    det = helpers.make_scanner(case["mode"], case["kit"])
    reads = helpers.case_reads(case, det.layouts)
    _, tr_striped, _ = oracle_lib.scan(det.descriptor(), reads, trace=True, rows=True)
    with helpers.r1_rule("scalar"):
        desc = det.descriptor()
    _, tr_scalar, _ = oracle_lib.scan(desc, reads, trace=True, rows=True)
    nt = len(det.layouts)
    differ = 0
    for a, b in zip(tr_striped, tr_scalar):
        assert list(a["tpl_raw"][:nt]) == list(b["tpl_raw"][:nt])
        differ += int(list(a["tpl_end"][:nt]) != list(b["tpl_end"][:nt]))
    assert differ > 0


def test_sg_r1_scalar_vs_independent_dp():
    """qo_sg_rule(QCAT_R1_SCALAR) against the independent scalar DP's rule="scalar" on seeded pairs (both matrices, ties forced by
    short and low-complexity sequences), and both rules agree on the score"""
    sys_path = os.path.join(helpers.GOLDEN)
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import sg_independent
    cfg = config.qcatConfig()
    rng = np.random.RandomState(20260930)
    for table, open_, ext, alpha in ((cfg.matrix.table, cfg.gap_open, cfg.gap_extend, "ACGTN"), (cfg.matrix_barcode.table, 1, 1, "ACGT")):
        score = sg_independent.scorer_from_table7(table)
        n_diff = 0
        for _ in range(400):
            k = int(rng.randint(2, 5))
            s1 = "".join(alpha[i] for i in rng.randint(0, k, size=int(rng.randint(1, 60))))
            s2 = "".join(alpha[i] for i in rng.randint(0, k, size=int(rng.randint(1, 50))))
            want = sg_independent.sg(s1, s2, open_, ext, score, rule="scalar")
            got = oracle_lib.sg(s1, s2, open_, ext, table, rule=native.R1_SCALAR)
            assert got == want, (s1, s2, got, want)
            striped = oracle_lib.sg(s1, s2, open_, ext, table)
            assert striped == sg_independent.sg(s1, s2, open_, ext, score)
            assert striped[0] == got[0]
            n_diff += int(striped != got)
        assert n_diff > 0
