"""Shared helpers of the test-suite: golden fixtures, scanner/descriptor construction and the
record comparison used for BOTH the oracle-vs-reference tests (CPU) and the HIP-vs-oracle /
HIP-vs-golden tests (GPU)."""
import json
import os

import numpy as np

import synth
from qcat_amd import config, native, scanner

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
_cache = {}


def golden():
    if "g" not in _cache:
        with open(os.path.join(GOLDEN, "golden_vectors.json")) as fh:
            _cache["g"] = json.load(fh)
    return _cache["g"]


def golden_r1_scalar():
    """the reference's outputs with plain `parasail.sg` bound (make_golden.py --r1-scalar; include/qcat_hip.h QCAT_R1_SCALAR)"""
    if "r1" not in _cache:
        with open(os.path.join(GOLDEN, "golden_r1_scalar.json")) as fh:
            _cache["r1"] = json.load(fh)
    return _cache["r1"]


class r1_rule(object):
    """with helpers.r1_rule("scalar"): ... -- descriptors built inside carry the rule (native.set_r1_rule)"""

    def __init__(self, rule):
        self.rule = rule

    def __enter__(self):
        self.old = native.set_r1_rule(self.rule)

    def __exit__(self, *exc):
        native.set_r1_rule(self.old)


def inline_reads():
    with open(os.path.join(GOLDEN, "inline_reads.json")) as fh:
        return json.load(fh)["reads"]


def fastq_records(name):
    with open(os.path.join(GOLDEN, "data", name)) as fh:
        lines = fh.read().split("\n")
    return [(lines[i][1:], lines[i + 1]) for i in range(0, len(lines) - 3, 4)]


def make_scanner(mode, kit, min_quality=None):
    return scanner.factory(mode=mode, kit=kit, min_quality=min_quality)


def case_reads(case, layouts):
    if "reads" in case:
        return case["reads"]
    g = case["gen"]
    return synth.synth_batch(g["n"], g["seed"], layouts, g["tpl_5p"], g["tpl_3p"],
                             error_rate=g["error_rate"], no_adapter_fraction=g["no_adapter_fraction"],
                             insert_len=g["insert_len"], lead_min=g["lead_min"], lead_max=g["lead_max"])


def record_as_golden(rec, layouts, mode):
    """native/oracle record -> the JSON shape make_golden.py stores for a reference dict."""
    out = {"barcode_id": None, "barcode_name": None, "adapter_kit": None,
           "adapter_idx": int(rec["adapter_idx"]), "adapter_end": int(rec["adapter_end"]),
           "trim5p": int(rec["trim5p"]), "trim3p": int(rec["trim3p"]),
           "exit_status": int(rec["exit_status"])}
    score = 0.0
    if rec["adapter_idx"] >= 0:
        out["adapter_kit"] = layouts[rec["adapter_idx"]].kit
    if rec["barcode_idx"] >= 0:
        lay = layouts[rec["adapter_idx"]]
        b1 = lay.get_barcode_set(0)[rec["barcode_idx"]]
        if mode == "dual":
            b2 = lay.get_barcode_set(1)[rec["barcode2_idx"]]
            out["barcode_id"] = "{}/{}".format(b1.id, b2.id)
            out["barcode_name"] = "barcode{:02d}/{:02d}".format(b1.id, b2.id)
        else:
            out["barcode_id"], out["barcode_name"] = b1.id, b1.name
        score = int(rec["raw_score"]) * 100.0 / (1.0 * int(rec["score_den"]))
    out["score_hex"] = float(score).hex()
    return out


def assert_case_matches(case, recs, traces, rows, layouts):
    """Compare records (+ optional traces / per-barcode rows) with one golden case."""
    mode = case["mode"]
    assert [l.kit for l in layouts] == case["layout_kits"]
    assert [l.get_adapter_length() for l in layouts] == case["layout_lens"]
    assert len(recs) == len(case["records"])
    for i, want in enumerate(case["records"]):
        got = record_as_golden(recs[i], layouts, mode)
        assert got == want["result"], "%s read %d: %r != %r" % (case["name"], i, got, want["result"])
        if traces is None:
            continue
        for e, wend in enumerate(want["ends"]):
            tr = traces[2 * i + e]
            nt = len(layouts)
            assert tr["window_len"] == wend["window_len"]
            if wend["tpl_raw"]:
                assert list(tr["tpl_raw"][:nt]) == wend["tpl_raw"], (case["name"], i, e)
                assert list(tr["tpl_end"][:nt]) == wend["tpl_end"], (case["name"], i, e)
            assert [int(tr["best_tpl"]), int(tr["best_end"])] == wend["best"][:2], (case["name"], i, e)
            # regions: the reference calls extract_barcode_region only on the region path
            sets_with_region = {s: n for s, n in wend["regions"]}
            for k, (widx, rlen, _ctx) in enumerate(wend["winners"]):
                if mode != "dual" and k == 1:
                    continue          # epi2me's discarded second scan (scanner_epi2me.py:104-131)
                assert int(tr["region_len"][k]) == rlen, (case["name"], i, e, k)
                assert int(tr["bc_idx"][k]) == widx, (case["name"], i, e, k)
                if k in sets_with_region:
                    assert sets_with_region[k] == rlen
            assert bool(tr["region_path"]) == (0 in sets_with_region)
            if rows is not None and wend.get("rows") is not None:
                for k, wrow in enumerate(wend["rows"]):
                    if mode != "dual" and k == 1:
                        continue
                    assert list(rows[2 * i + e, k, :len(wrow)]) == wrow, (case["name"], i, e, k)


def middle_reads(entry, layouts):
    """Re-create the read list of one `middle` fixture (see make_golden.py section 8)."""
    g = entry["gen"]
    base = synth.synth_batch(48, g["seed"], layouts, g["tpl_5p"], g["tpl_3p"], error_rate=g["error_rate"])
    reads = []
    for i in range(24):
        if i % 3 == 0:
            reads.append(base[i] + base[i + 24])
        elif i % 3 == 1:
            reads.append(base[i])
        else:
            reads.append(base[i][:120 + 17 * i])
    return reads


def long_scan_inputs(entry, layouts):
    """Re-create the sequences of one `long_scan` fixture (see make_golden.py section 9):
    -> (sequences for scan(), reads for scan_middle())."""
    g = entry["gen"]
    base = synth.synth_batch(12, g["seed"], layouts, g["tpl_5p"], g["tpl_3p"], error_rate=g["error_rate"])
    seqs = []
    for i, r in enumerate(base):
        cut = (151, 152, 200, 299, 300, 301, 450, 640, len(r), len(r), len(r), len(r))[i]
        s = r[:cut]
        if i >= 10:
            s = s[200:] + s[:200]
        seqs.append(s)
    chim = [base[i] + base[i + 6] for i in range(6)] + base[:3] + [base[0][:300], base[1][:301], ""]
    return seqs, chim


def driver_histogram(desc, layouts, recs, read_lengths, min_read_length, trim):
    """The count vector the reference DRIVER would accumulate for these records (qcat/cli.py:515-534 +
    :366-383), restated with Python slicing: trim, drop reads under the min-length filter into
    `skipped`, count the rest by barcode and by kit.  Layout: [barcodes.., none][kits.., none][skipped]."""
    nb, nk = len(desc.slot_ids), len(desc.kit_names)
    dual = desc.mode == "dual"
    nbc = nb * nb if dual else nb
    cnt = np.zeros(desc.n_count_buckets, dtype=np.int64)
    for rec, n in zip(recs, read_lengths):
        sequence = "x" * int(n)
        if trim:
            sequence = sequence[int(rec["trim5p"]):int(rec["trim3p"])]
        if len(sequence) < min_read_length:
            cnt[-1] += 1
            continue
        slot = nbc
        if rec["barcode_idx"] >= 0:
            lay = layouts[rec["adapter_idx"]]
            slot = desc.id_slots[lay.get_barcode_set(0)[rec["barcode_idx"]].id]
            if dual:
                slot = slot * nb + desc.id_slots[lay.get_barcode_set(1)[rec["barcode2_idx"]].id]
        cnt[slot] += 1
        cnt[nbc + 1 + (desc.kit_slots[layouts[rec["adapter_idx"]].kit] if rec["adapter_idx"] >= 0 else nk)] += 1
    return cnt


def median_kept_length(det, reads, trim):
    """median length of the (trimmed) reads under the oracle: a min-length threshold that splits the batch"""
    import oracle_lib
    recs = oracle_lib.scan(det.descriptor(), reads, threads=8)
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    if trim:
        lens = np.maximum(np.minimum(recs["trim3p"], lens) - np.minimum(recs["trim5p"], lens), 0)
    return int(np.median(lens))


def simple_reads(entry):
    """Re-create the reads of one `simple` fixture (make_golden.py section 10)."""
    g = entry["gen"]
    lays = scanner.factory(mode="epi2me", kit=g["kit"]).layouts
    return synth.synth_batch(g["n"], g["seed"], lays, g["tpl_5p"], g["tpl_3p"], error_rate=g["error_rate"]) + entry["extra"]


def simple_record_as_golden(rec, barcodes):
    b = int(rec["barcode_idx"])
    score = int(rec["raw_score"]) * 100.0 / (1.0 * int(rec["score_den"])) if b >= 0 else 0.0
    return {"barcode_index": b, "barcode_name": barcodes[b].name if b >= 0 else None,
            "barcode_id": barcodes[b].id if b >= 0 else None, "score_hex": float(score).hex(),
            "adapter": None if rec["adapter_idx"] < 0 else "?", "adapter_end": int(rec["adapter_end"]),
            "trim5p": int(rec["trim5p"]), "trim3p": int(rec["trim3p"]), "exit_status": int(rec["exit_status"])}
