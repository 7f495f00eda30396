"""Host-side mirror of the reference interface (CPU only).  The first four tests carry the same
assertions as the reference's own unit tests for these helpers (qcat/test/test_barcode.py:15-67
placeholder positions, :426-435 phred helpers, :438-477 window extraction, :492-544 adapter
layout); the rest pin the registry, kit selection, config quirks and the descriptor flattening."""
import os

import numpy as np
import pytest

from qcat_amd import adapters, config, native, scanner, utils
from qcat_amd.scanner_base import extract_align_sequence

SPACER = "N" * 24


def test_get_placeholder():
    gp = adapters.AdapterLayout.get_placeholder_pos
    assert tuple(gp("NNNNN")) == (0, 4, 5)
    assert tuple(gp("AAAANNNNN")) == (4, 8, 5)
    assert tuple(gp("NNNNNAAAA")) == (0, 4, 5)
    assert tuple(gp("")) == (-1, -1, 0)
    plain = "AATGTACTTCGTTCAGTTACGTATTGCTGTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"
    assert tuple(gp(plain)) == (-1, -1, 0)
    assert tuple(gp(plain[:28] + "N" + plain[28:])) == (28, 28, 1)
    assert tuple(gp(plain[:28] + SPACER + plain[28:])) == (28, 51, 24)
    two = plain[:28] + SPACER + "GTTTTCGCATTTATCGTG" + "N" * 10 + "AAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"
    assert tuple(gp(two, 0)) == (28, 51, 24)
    assert tuple(gp(two, 1)) == (70, 79, 10)


def test_utils():
    assert utils.qstring_to_phred(None) == []
    assert utils.qstring_to_phred("") == []
    assert utils.qstring_to_phred("IIII") == [40, 40, 40, 40]
    assert utils.mean_error_prob(None) == -1
    assert utils.mean_error_prob([]) == -1
    assert utils.mean_error_prob([40]) == 0.0001
    assert utils.mean_error_prob([40, 40, 40, 40]) == 0.0001
    assert utils.mean_error_prob([40, 20, 20, 40]) == 0.00505
    assert utils.revcomp("ACGTNacgtRYKMx-") == "-xKMRYacgtNACGT"


def test_extract_align_sequence():
    for seq in (None, ""):
        for rev in (True, False):
            for n in (100, 0, -1):
                assert extract_align_sequence(seq, rev, n) == ""
    read = "AGTATTACTTCGTTCAGTTACGTATTGCTGTTTCATCTATCAGGAGGGAATGGAGTTTCGC"
    assert extract_align_sequence(read, False, 10) == "AGTATTACTT"
    assert extract_align_sequence(read, True, 10) == "GCGAAACTCC"
    for rev in (True, False):
        for n in (0, -1):
            assert extract_align_sequence(read, rev, n) == read


def test_adapter_layout():
    simple = adapters.get_barcodes_simple()
    assert len(simple) == 24 and len(adapters.get_barcodes_simple("extended")) == 120
    head, tail = "AAAAAAAAAT", "ATTTTTTTTTGGGGGGGGGC"
    seq = head + SPACER + tail + SPACER + "GCCCCCCCCC"
    lay = adapters.AdapterLayout("PBC001", seq, simple, simple, "Test layout")
    assert lay.get_adapter_sequences() == seq and lay.get_adapter_length() == len(seq)
    assert lay.get_barcode_end(0) == len(head) + 24 - 1
    assert lay.get_barcode_end(1) == len(head + SPACER + tail + SPACER) - 1
    assert lay.get_barcode_length(0) == 24 and lay.get_barcode_length(1) == 24
    assert lay.get_barcode_set(0) == simple and lay.get_barcode_set(1) == simple
    assert lay.get_downstream_context(2, 0) == "AT" and lay.get_downstream_context(2, 1) == "GC"
    assert lay.get_upstream_context(2, 0) == "AT" and lay.get_upstream_context(2, 1) == "GC"
    assert lay.is_double_barcode()
    single = adapters.AdapterLayout("PBC001", head + SPACER + "ATTTTTTTTT", simple, None, "x")
    assert not single.is_double_barcode() and single.get_barcode_length(1) == 0 and single.get_barcode_end(1) == -1
    assert single.get_upstream_context(11, 0) == head and single.get_upstream_context(3, 1) == ""
    assert single.get_adapter_sequences("ACGT") == head + "ACGT" + "ATTTTTTTTT"
    with pytest.raises(RuntimeError):
        adapters.AdapterLayout("K", "ACGU", simple, None, "bad character")
    with pytest.raises(RuntimeError):
        adapters.AdapterLayout("K", head + "N" * 20 + tail, simple, None, "placeholder length mismatch")
    with pytest.raises(RuntimeError):
        single.get_barcode_end(2)


def test_registry_and_kit_selection():
    assert scanner.get_modes() == ["epi2me", "dual", "simple"]
    kits = scanner.get_kits()
    assert kits[0] == "Auto" and "PBC096" in kits and len(kits) == 15
    assert scanner.get_kits_info()["PBC096"] == "PCR Barcoding Kit with 96 barcodes"
    assert [l.get_adapter_length() for l in scanner.get_adapter_by_name("PBC096")] == [60, 59]     # sorted: 3p, 5p
    assert len(scanner.factory().layouts) == 12                                  # auto_detect templates
    assert len(scanner.factory(kit="auto").layouts) == 12
    assert [l.kit for l in scanner.factory(kit="pbc096").layouts] == ["PBC096", "PBC096"]   # case-insensitive
    assert scanner.factory(kit="no-such-kit").layouts == []
    assert scanner.factory().min_quality == 58 and scanner.factory(mode="dual").min_quality == 60
    assert scanner.factory(min_quality=71.5).min_quality == 71.5
    assert [l.kit for l in scanner.factory(mode="dual", kit="PBC096").layouts] == ["DUAL", "DUAL"]   # kit is ignored
    assert scanner.factory(mode="guppy").get_name() == "epi2me"
    with pytest.raises(RuntimeError, match="Invalid demultiplexing mode"):
        scanner.factory(mode="brill")
    simple = scanner.factory(mode="simple", kit="standard")
    assert simple.get_name() == "simple" and simple.min_quality == 60 and len(simple.barcodes) == 24
    assert simple.barcode_count() == 25 and len(scanner.factory(mode="simple", kit="extended").barcodes) == 120


def test_config_quirks_and_ini_roundtrip(tmp_path):
    c = config.qcatConfig()
    assert (c.match, c.nmatch, c.mismatch, c.gap_open, c.gap_extend) == (5, -1, -2, 2, 2)
    assert (c.max_align_length, c.extracted_barcode_extension, c.barcode_context_length) == (150, 11, 11)
    c.mismatch = 3
    assert c.mismatch == -3 and c.matrix.score("A", "C") == -3
    c.match = -4
    assert c.match == 4 and c.matrix.score("G", "g") == 4
    c.nmatch = -2                      # the reference's setter stores abs(): N then scores +2
    assert c.nmatch == 2 and c.matrix.score("A", "N") == 2
    c.gap_open = -5
    assert c.gap_open == 5
    path = str(tmp_path / "qcat.ini")
    c.max_align_length = 120
    c.write(path)
    d = config.qcatConfig(path)
    assert (d.match, d.mismatch, d.gap_open, d.gap_extend, d.max_align_length) == (4, -3, 5, 2, 120)
    assert config.get_default_config().fingerprint() == config.qcatConfig().fingerprint()


def test_descriptor_flattening():
    det = scanner.factory(mode="dual")
    d = det.descriptor()
    assert d.desc.n_templates == 2 and d.desc.mode == native.MODE_DUAL and d.desc.min_quality == 60.0
    t = d.desc.templates[0]
    assert (t.length, t.is_double_barcode, t.bc_start[0], t.bc_end[0], t.bc_start[1], t.bc_end[1]) == (83, 1, 6, 29, 45, 68)
    assert (t.sets[0].n, t.sets[1].n, t.sets[0].barcode_len) == (24, 96, 24)
    assert d.desc.n_barcode_slots == 96 and d.n_count_buckets == 96 * 96 + 1 + 1 + 1 + 1
    e = scanner.factory(kit="RPB004/RLB001").descriptor()           # id 12 appears twice (barcode12, barcode12a)
    ids = [e.desc.templates[0].sets[0].ids[i] for i in range(13)]
    assert ids[11] == ids[12] and len(set(ids)) == 12
    bases, offsets = native.pack_reads(["ACGT", None, "", "TT"])
    assert list(offsets) == [0, 4, 4, 4, 6] and bases.tobytes() == b"ACGTTT"


def test_kit_folder_yaml_loader(tmp_path):
    import yaml
    data = {"kit": "MYKIT", "auto_detect": True, "description": "d", "sequence": "ACGT" + SPACER + "TTGCA", "trim_offset": 3,
            "barcode_set_1": [{"name": "barcode01", "id": 1, "sequence": "A" * 24, "fwd_strand": True}], "barcode_set_2": []}
    with open(os.path.join(str(tmp_path), "b.yml"), "w") as fh:
        yaml.safe_dump(data, fh)
    data["kit"] = "OTHER"
    data["active"] = False
    with open(os.path.join(str(tmp_path), "a.yml"), "w") as fh:
        yaml.safe_dump(data, fh)
    lays = adapters.populate_adapter_layouts(str(tmp_path))
    assert [l.kit for l in lays] == ["MYKIT"]                       # inactive entries are skipped
    assert lays[0].trim_offset == 3 and lays[0].barcode_set_2 is None and not lays[0].is_double_barcode()
    assert scanner.factory(kit_folder=str(tmp_path)).layouts[0].kit == "MYKIT"        # auto_detect honoured


def test_fastx_parsers_follow_biopython(tmp_path):
    """the reference reads FASTQ with Bio's FastqGeneralIterator and FASTA with SimpleFastaParser
    (qcat/cli.py:260,287): wrapped records, stripped titles, blanks inside FASTA sequences, and
    status 1 on a malformed record."""
    import io
    from qcat_amd import cli
    plain = "@r1 some comment  \nACGT\n+\nIIII\n@r2\nGG\n+r2\n@I\n\n"
    assert list(cli._fastq_records(io.StringIO(plain))) == [("r1 some comment", "ACGT", "IIII"), ("r2", "GG", "@I")]
    wrapped = "\n@w1 c\t\nACGT\nAC\n+\nIII\n@II\n@w2\nTT\n+\nII\n"
    assert list(cli._fastq_records(io.StringIO(wrapped))) == [("w1 c", "ACGTAC", "III@II"), ("w2", "TT", "II")]
    mixed = "@a\nAC\n+\nII\n@b\nACG\nT\n+\nIIII\n@c\nA\n+\nI\n"                    # the fast path hands over mid-file
    assert [r[0] for r in cli._fastq_records(io.StringIO(mixed))] == ["a", "b", "c"]
    for bad, msg in (("@x\nACGT\n+\nII\n", "Lengths of sequence and quality"), ("x\nACGT\n+\nIIII\n", "should start with '@'"),
                     ("@x\nAC GT\n+\nIIIII\n", "Whitespace"), ("@x\nACGT\n+y\nIIII\n", "captions differ"), ("@x\nACGT\n", "without quality")):
        with pytest.raises(ValueError, match=msg):
            list(cli._fastq_records(io.StringIO(bad)))
    fasta = "junk before\n>t1 desc  \nAC GT\r\nTT\n>t2\n\n>t3\nA\n"
    assert list(cli._fasta_records(io.StringIO(fasta))) == [("t1 desc", "ACGTTT"), ("t2", ""), ("t3", "A")]
    bad = tmp_path / "bad.fastq"
    bad.write_text("@ok\nAC\n+\nII\n@broken\nACGT\n+\nII\n")
    with pytest.raises(SystemExit) as exc:
        list(cli.iter_fastx(str(bad), True, 10))
    assert exc.value.code == 1


def test_the_c_helper_builds_the_same_dicts_and_hands_over_the_strings_uncopied():
    """qcat_amd/_pyglue.so (csrc/pyglue.c): `records_to_dicts` must give exactly the dicts of the Python loop -- same
    objects from the kit's tables, the same IEEE double for the score -- and `read_views` one (pointer, length) per str /
    bytes / None, or None for anything it does not take (the caller then packs the list as before).  Dual records (a
    Barcode per pair of ids) stay on the Python loop."""
    import ctypes
    import struct
    from qcat_amd import native, scanner
    if native._pyglue is None:
        pytest.skip("the optional C helper is not built")
    det = scanner.factory(mode="epi2me", kit=None)
    rng = np.random.RandomState(5)
    n = 500
    recs = np.zeros(n, dtype=native.RESULT_DTYPE)
    recs["adapter_idx"] = rng.randint(-1, len(det.layouts), n)
    nb = np.array([len(l.get_barcode_set(0) or ()) for l in det.layouts])
    recs["barcode_idx"] = np.where(recs["adapter_idx"] >= 0, rng.randint(-1, 12, n) % np.maximum(nb[np.maximum(recs["adapter_idx"], 0)], 1), -1)
    recs["barcode_idx"][::7] = -1
    recs["barcode2_idx"] = -1
    recs["raw_score"] = rng.randint(-5, 47, n); recs["score_den"] = rng.choice([39, 41, 42, 45, 46], n)
    recs["adapter_end"] = rng.randint(0, 150, n); recs["trim5p"] = rng.randint(0, 150, n); recs["trim3p"] = rng.randint(0, 9000, n)
    recs["exit_status"] = rng.choice([0, 1, 1002], n)
    with_c = det._records_to_dicts(recs, det.layouts)
    glue, native._pyglue = native._pyglue, None
    try:
        det._dict_tables = None
        plain = det._records_to_dicts(recs, det.layouts)
    finally:
        native._pyglue = glue
    assert with_c == plain and len(with_c) == n
    for a, b in zip(with_c, plain):
        assert a["barcode"] is b["barcode"] and a["adapter"] is b["adapter"] and list(a) == list(b)
        assert float(a["barcode_score"]).hex() == float(b["barcode_score"]).hex()
    recs["barcode2_idx"][3] = 2                                   # a dual record: not the helper's business
    assert glue.records_to_dicts(np.ascontiguousarray(recs), [[None]], [None]) is None
    # read_views: the objects' own buffers
    reads = ["ACGT" * 5, b"TTTT", None, ""]
    ptrs, lens = native.read_views(reads)
    p = struct.unpack("<4Q", ptrs)
    assert struct.unpack("<4Q", lens) == (20, 4, 0, 0) and p[2] == 0
    assert ctypes.string_at(p[0], 20) == b"ACGT" * 5 and ctypes.string_at(p[1], 4) == b"TTTT"
    assert native.read_views(["ACGT", "é"]) is None and native.read_views(["ACGT", 7]) is None and native.read_views(("A",)) is None
