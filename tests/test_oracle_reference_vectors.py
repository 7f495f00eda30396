"""Pin the oracle against every known answer the reference's own tests hold for the hot path
(SURVEY.md section 8c).  CPU only."""
import pytest

import helpers
import oracle_lib
from qcat_amd import config, native, scanner


def _names(det, reads, **kw):
    d = det.descriptor()
    recs = oracle_lib.scan(d, reads)
    return [helpers.record_as_golden(r, det.layouts, det._native_mode)["barcode_name"] for r in recs]


def test_find_best_adapter_template_known_answer():
    """qcat/test/test_barcode.py:291-304: RBK001 x read_bc3_exact -> template 0, end_query 101."""
    det = scanner.factory(kit="RBK001")
    d = det.descriptor()
    _, tr = oracle_lib.scan(d, [helpers.inline_reads()["read_bc3_exact"]], trace=True)
    assert tr[0]["best_tpl"] == 0
    assert tr[0]["best_end"] == 101
    # (None, None) -> (-1, -1): empty window
    _, tr = oracle_lib.scan(d, [""], trace=True)
    assert tr[0]["best_tpl"] == -1 and tr[0]["best_end"] == -1


def test_barcode_kit_auto():
    """test_barcode.py:70-85 and :352-400 (kit auto)."""
    r = helpers.inline_reads()
    det = scanner.factory()
    got = _names(det, [r["read"], r["read_bc3_exact"], r["read_bc3"], r["real_bc03_porechop"],
                       r["read_nobc"], ""])
    assert got == ["barcode02", "barcode03", "barcode03", "barcode03", None, None]


def test_barcode_rapidkit():
    """test_barcode.py:307-322 (kit RBK001)."""
    r = helpers.inline_reads()
    det = scanner.factory(kit="RBK001")
    got = _names(det, [r["read"], r["read_bc3_exact"], r["read_bc3"], r["real_bc03_porechop"],
                       r["read_nobc"], ""])
    assert got == ["barcode02", "barcode03", "barcode03", "barcode03", None, None]


@pytest.mark.parametrize("fname", ["nbd103.fastq", "pbk004.fastq", "rab204.fastq", "rbk004.fastq"])
def test_full_run(fname):
    """test_barcode.py:556-604, :617-654: kit auto, every call equals truebc, trims sane."""
    recs_in = helpers.fastq_records(fname)
    det = scanner.factory()
    recs = oracle_lib.scan(det.descriptor(), [s for _, s in recs_in])
    for (head, seq), rec in zip(recs_in, recs):
        true_bc = [t for t in head.split() if t.startswith("truebc=")][0].split("=")[1]
        got = helpers.record_as_golden(rec, det.layouts, "epi2me")
        assert str(got["barcode_id"]) == true_bc, head
        assert rec["trim5p"] < 200
        assert len(seq) - rec["trim3p"] < 200
        assert len(seq) >= rec["trim3p"] - rec["trim5p"]


def test_matrix_values():
    """qcat/config.py:26, :236-253 as dumped through the reference."""
    cfg = config.qcatConfig()
    m = cfg.matrix
    assert m.score("A", "A") == 5 and m.score("A", "C") == -2 and m.score("a", "A") == 5
    assert m.score("A", "N") == -1 and m.score("N", "N") == -1
    assert m.score("A", "X") == 0 and m.score("X", "X") == 0 and m.score("A", "R") == 0
    b = cfg.matrix_barcode
    assert b.score("A", "A") == 1 and b.score("N", "N") == 1 and b.score("A", "N") == -1
    assert b.score("A", "X") == 0 and b.score("U", "A") == 0
