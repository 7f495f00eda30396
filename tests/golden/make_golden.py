#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the UNMODIFIED reference Python
(/root/reference, read-only) in the authoring container.  Dev tool: it cannot run on the
GPU box (no /root/reference there) and nothing in tests/ calls it.

What is real and what is a stand-in
-----------------------------------
The reference imports two third-party packages that are absent from this image:
``parasail`` (all DP arithmetic) and ``Bio`` (FASTA/FASTQ parsing).  This script puts two
minimal stand-in modules on ``sys.path`` (written to a temp dir, never committed as part of
the product):
  * ``parasail``: ``matrix_create`` (mutable ``pointer[0].matrix`` as qcat/config.py:247-253
    needs), ``sg_striped_32``/``sg`` computed by the INDEPENDENT scalar Python DP of
    tests/golden/sg_independent.py (written from the published recurrence; it does not call, link
    or share code with oracle/qcat_oracle.c), ``can_use_sse2``.
  * ``Bio.SeqIO.FastaIO.SimpleFastaParser`` / ``Bio.SeqIO.QualityIO.FastqGeneralIterator``.
Consequence: the fixtures pin everything ABOVE the parasail call -- windowing, template
arg-max, region slicing, the barcode arg-max quirk, thresholds, trims, conflict handling,
dual combination, batch/kit vote -- against the reference's own code, and every DP value in them
(per-template raw score and end_query, every per-barcode raw score) is one the oracle did not
produce: tests/test_oracle_golden.py checks the oracle's C DP against them.  What remains
unpinned is only whether real parasail agrees with the published recurrence + rule R1 beyond the
reference's known answers (tests/test_oracle_reference_vectors.py): parasail is not available.
The oracle is NOT imported by this script.

Template order: ``glob.glob`` is wrapped with ``sorted`` before qcat is imported, i.e. the
sorted-by-file-name order this build fixes.

Outputs (tests/golden/):
  inline_reads.json      the inline read strings held by qcat/test/test_barcode.py (data)
  data/*.fastq           the four FASTQ files of qcat/test/data (data)
  golden_vectors.json    reference outputs for the cases below
"""
import glob
import json
import os
import shutil
import sys
import tempfile
import textwrap

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

sys.path.insert(0, HERE)

import sg_independent  # noqa: E402,F401  (imported by the parasail stand-in below)
import synth  # noqa: E402


def install_standins():
    d = tempfile.mkdtemp(prefix="qcat_standins_")
    os.makedirs(os.path.join(d, "Bio", "SeqIO"))
    open(os.path.join(d, "Bio", "__init__.py"), "w").close()
    open(os.path.join(d, "Bio", "SeqIO", "__init__.py"), "w").close()
    with open(os.path.join(d, "Bio", "SeqIO", "FastaIO.py"), "w") as fh:
        fh.write(textwrap.dedent('''
            def SimpleFastaParser(handle):
                title, seq = None, []
                for line in handle:
                    line = line.rstrip("\\n")
                    if line.startswith(">"):
                        if title is not None:
                            yield title, "".join(seq)
                        title, seq = line[1:], []
                    elif title is not None:
                        seq.append(line.strip())
                if title is not None:
                    yield title, "".join(seq)
            '''))
    with open(os.path.join(d, "Bio", "SeqIO", "QualityIO.py"), "w") as fh:
        fh.write(textwrap.dedent('''
            def FastqGeneralIterator(handle):
                while True:
                    head = handle.readline()
                    if not head:
                        return
                    seq = handle.readline().rstrip("\\n")
                    handle.readline()
                    qual = handle.readline().rstrip("\\n")
                    yield head.rstrip("\\n")[1:], seq, qual
            '''))
    with open(os.path.join(d, "parasail.py"), "w") as fh:
        fh.write(textwrap.dedent('''
            """Stand-in for parasail backed by the independent scalar DP tests/golden/sg_independent.py
            (see tests/golden/make_golden.py)."""
            import sg_independent

            CALLS = {"n": 0, "cells": 0, "computed": 0}
            _MEMO = {}

            class _Inner(object):
                def __init__(self, flat):
                    self.matrix = flat

            class Matrix(object):
                def __init__(self, alphabet, match, mismatch):
                    n = len(alphabet)
                    self.alphabet = alphabet
                    self.size = n + 1
                    flat = []
                    for i in range(n):
                        flat += [match if i == j else mismatch for j in range(n)] + [0]
                    flat += [0] * (n + 1)
                    self.pointer = [_Inner(flat)]

                def scorer(self):
                    """W(a, b) through the mapper: alphabet letters in either case, the rest -> '*'."""
                    flat = tuple(self.pointer[0].matrix)
                    return flat, sg_independent.make_scorer(self.alphabet, flat, self.size)

            class Result(object):
                def __init__(self, score, end_query, end_ref):
                    self.score, self.end_query, self.end_ref = score, end_query, end_ref

            import os
            NO_SSE2 = os.environ.get("QCAT_GOLDEN_NO_SSE2") == "1"      # make_r1_golden.py: the reference then binds plain `sg`

            def can_use_sse2():
                return not NO_SSE2

            def matrix_create(alphabet, match, mismatch):
                return Matrix(alphabet, match, mismatch)

            def sg_striped_32(s1, s2, open, extend, matrix):
                CALLS["n"] += 1
                CALLS["cells"] += len(s1) * len(s2)
                flat, score = matrix.scorer()
                key = (s1, s2, open, extend, matrix.alphabet, flat)
                if key not in _MEMO:
                    CALLS["computed"] += 1
                    _MEMO[key] = sg_independent.sg(s1, s2, open, extend, score)
                return Result(*_MEMO[key])

            def sg(s1, s2, open, extend, matrix):
                """plain parasail.sg: the same DP, the scalar routine's end-position order (sg_independent.sg rule="scalar")"""
                CALLS["n"] += 1
                CALLS["cells"] += len(s1) * len(s2)
                flat, score = matrix.scorer()
                key = ("scalar", s1, s2, open, extend, matrix.alphabet, flat)
                if key not in _MEMO:
                    CALLS["computed"] += 1
                    _MEMO[key] = sg_independent.sg(s1, s2, open, extend, score, rule="scalar")
                return Result(*_MEMO[key])

            class StatsResult(Result):
                def __init__(self, score, end_query, end_ref, matches, length):
                    Result.__init__(self, score, end_query, end_ref)
                    self.matches, self.length = matches, length

            def sg_stats_striped_32(s1, s2, open, extend, matrix):
                """simple mode (scanner_simple.py:70-74): score / end_query as sg; matches / length for completeness
                (nothing on the scanner paths consumes them, see sg_independent.sg_stats)"""
                CALLS["n"] += 1
                CALLS["cells"] += len(s1) * len(s2)
                flat, score = matrix.scorer()
                key = ("stats", s1, s2, open, extend, matrix.alphabet, flat)
                if key not in _MEMO:
                    _MEMO[key] = sg_independent.sg_stats(s1, s2, open, extend, score, alphabet=matrix.alphabet)
                return StatsResult(*_MEMO[key])

            def sg_stats(s1, s2, open, extend, matrix):
                CALLS["n"] += 1
                CALLS["cells"] += len(s1) * len(s2)
                flat, score = matrix.scorer()
                key = ("stats-scalar", s1, s2, open, extend, matrix.alphabet, flat)
                if key not in _MEMO:
                    _MEMO[key] = sg_independent.sg_stats(s1, s2, open, extend, score, alphabet=matrix.alphabet, r1="scalar")
                return StatsResult(*_MEMO[key])
            '''))
    sys.path.insert(0, d)
    return d


def import_reference():
    _glob = glob.glob
    glob.glob = lambda *a, **k: sorted(_glob(*a, **k))        # R8: sorted template order
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import qcat.scanner as ref_scanner                        # noqa
    import qcat.scanner_base as ref_base                      # noqa
    import qcat.scanner_epi2me as ref_epi                     # noqa
    import qcat.scanner_dual as ref_dual                      # noqa
    import qcat.config as ref_config                          # noqa
    return ref_scanner, ref_base, ref_epi, ref_dual, ref_config


class Tracer(object):
    """Records the intermediates of every scan() by wrapping the helper functions in the
    namespaces of the two scanner modules (they import the helpers by name)."""

    def __init__(self, mods):
        self.ends = []
        self.cur = None
        for mod in mods:
            self._wrap(mod)

    def _wrap(self, mod):
        fbt, ebr, fhs = mod.find_best_adapter_template, mod.extract_barcode_region, mod.find_highest_scoring_barcode
        tracer = self
        import parasail

        def find_best(adapter_templates, read_sequence, qcat_config):
            tracer.cur = {"window_len": len(read_sequence or ""), "tpl_raw": [], "tpl_end": [],
                          "regions": [], "rows": [], "winners": []}
            tracer.ends.append(tracer.cur)
            # per-template raw scores through the reference's own eval_adapter_template
            import qcat.scanner_base as sb
            if adapter_templates and read_sequence:
                for tpl in adapter_templates:
                    end, _, raw = sb.eval_adapter_template(tpl, read_sequence, qcat_config, identity=False)
                    tracer.cur["tpl_raw"].append(raw)
                    tracer.cur["tpl_end"].append(end)
            ret = fbt(adapter_templates=adapter_templates, read_sequence=read_sequence, qcat_config=qcat_config)
            tracer.cur["best"] = [ret[0], ret[1], float(ret[2]).hex()]
            return ret

        def region(read_sequence, adapter_template, barcode_set_index, alignment_stop_ref, qcat_config):
            out = ebr(read_sequence=read_sequence, adapter_template=adapter_template,
                      barcode_set_index=barcode_set_index, alignment_stop_ref=alignment_stop_ref,
                      qcat_config=qcat_config)
            tracer.cur["regions"].append([barcode_set_index, out])
            return out

        def highest(barcode_region_read, barcode_set, qcat_config, upstream_context="", downstream_context="",
                    compute_identity=False):
            row = []
            if barcode_region_read:
                for bc in barcode_set:
                    r = parasail.sg_striped_32(barcode_region_read, upstream_context + bc.sequence + downstream_context,
                                               1, 1, qcat_config.matrix_barcode)
                    row.append(r.score)
            ret = fhs(barcode_region_read=barcode_region_read, barcode_set=barcode_set, qcat_config=qcat_config,
                      upstream_context=upstream_context, downstream_context=downstream_context)
            tracer.cur["rows"].append(row)
            tracer.cur["winners"].append([barcode_set.index(ret[0]) if ret[0] is not None else -1,
                                          len(barcode_region_read or ""),
                                          len(upstream_context) + len(downstream_context)])
            return ret

        mod.find_best_adapter_template = find_best
        mod.extract_barcode_region = region
        mod.find_highest_scoring_barcode = highest

    def take(self):
        ends, self.ends = self.ends, []
        return ends


def result_to_json(res, layouts):
    bc = res["barcode"]
    ad = res["adapter"]
    return {"barcode_id": None if bc is None else bc.id,
            "barcode_name": None if bc is None else bc.name,
            "score_hex": float(res["barcode_score"]).hex(),
            "adapter_kit": None if ad is None else ad.kit,
            "adapter_idx": -1 if ad is None else [id(l) for l in layouts].index(id(ad)),
            "adapter_end": res["adapter_end"], "trim5p": res["trim5p"], "trim3p": res["trim3p"],
            "exit_status": res["exit_status"]}


def end_to_json(e, keep_rows):
    out = {"window_len": e["window_len"], "tpl_raw": e["tpl_raw"], "tpl_end": e["tpl_end"],
           "best": e["best"], "regions": [[s, len(r)] for s, r in e["regions"]],
           "region_text": [r for _, r in e["regions"]] if keep_rows else None,
           "winners": e["winners"]}
    if keep_rows:
        out["rows"] = e["rows"]
    return out


def simple_section(ref_scanner, cfg, fastq, inline, inline_names, seed0):
    """10. BarcodeScannerSimple (qcat/scanner_simple.py): detect_barcode with the bundled barcode lists over reads
    that carry PCR barcodes (the lists are the PCR barcode family, both strands)."""
    out = []
    lwb = ref_scanner.factory(mode="epi2me", kit="PBK004/LWB001").layouts
    pbc = ref_scanner.factory(mode="epi2me", kit="PBC096").layouts
    for which, lays, n in (("standard", lwb, 40), ("extended", pbc, 24)):
        det = ref_scanner.factory(mode="simple", kit=which)
        for e in (0.0, 0.1):
            gen = {"seed": seed0 + 40, "n": n, "tpl_5p": 1, "tpl_3p": 0, "error_rate": e, "kit": lays[0].kit}
            reads = synth.synth_batch(n, gen["seed"], lays, 1, 0, error_rate=e)
            extra = ["", "A", reads[0][:100], reads[1][:151], "N" * 200, reads[2].lower()]
            res = []
            for r in reads + extra:
                d = det.detect_barcode(r, qcat_config=cfg)
                bc = d["barcode"]
                res.append({"barcode_index": -1 if bc is None else det.barcodes.index(bc),
                            "barcode_name": None if bc is None else bc.name, "barcode_id": None if bc is None else bc.id,
                            "score_hex": float(d["barcode_score"]).hex(), "adapter": d["adapter"],
                            "adapter_end": d["adapter_end"], "trim5p": d["trim5p"], "trim3p": d["trim3p"],
                            "exit_status": d["exit_status"]})
            out.append({"list": which, "n_barcodes": len(det.barcodes), "min_quality": det.min_quality, "gen": gen,
                        "extra": extra, "results": res})
            print("simple/%s e=%.2f: %d reads, %d called, exits %s" % (which, e, len(res), sum(1 for x in res if x["barcode_name"]),
                                                                      sorted(set(x["exit_status"] for x in res))))
    return out


def main():
    # --r1-scalar (round 6): the stand-in reports no SSE2, so the UNMODIFIED reference binds plain `parasail.sg`
    # (qcat/scanner_base.py:20-26) -- the other end-position rule (include/qcat_hip.h QCAT_R1_SCALAR).  A smaller set of
    # cases goes to golden_r1_scalar.json: the shipped reads, the inline reads, the edge cases, synthetic reads of four
    # kits and adapter-FREE reads of two (where the two rules part: the borders' maxima tie in low-scoring windows).
    r1_scalar = "--r1-scalar" in sys.argv
    if r1_scalar:
        os.environ["QCAT_GOLDEN_NO_SSE2"] = "1"
    install_standins()
    if "--section" in sys.argv and sys.argv[sys.argv.index("--section") + 1] == "simple":
        ref_scanner, ref_base, ref_epi, ref_dual, ref_config = import_reference()
        path = os.path.join(HERE, "golden_vectors.json")
        with open(path) as fh:
            doc = json.load(fh)
        doc["simple"] = simple_section(ref_scanner, ref_config.qcatConfig(), None, None, None, 20260928)
        with open(path, "w") as fh:
            json.dump(doc, fh, separators=(",", ":"))
        print("golden_vectors.json:", os.path.getsize(path), "bytes (section simple regenerated)")
        return
    ref_scanner, ref_base, ref_epi, ref_dual, ref_config = import_reference()
    import parasail
    tracer = Tracer([ref_epi, ref_dual])
    cfg = ref_config.qcatConfig()

    # ---- data held by the reference's tests -------------------------------------------------
    import qcat.test.test_barcode as ref_tests
    inline_names = ["read", "read_bc3_exact", "read_bc3", "real_bc03_porechop", "read_nobc",
                    "real_double_barcode_read"]
    inline = {n: getattr(ref_tests, n) for n in inline_names if hasattr(ref_tests, n)}
    with open(os.path.join(HERE, "inline_reads.json"), "w") as fh:
        json.dump({"source": "qcat/test/test_barcode.py:71-288 (module-level read strings)", "reads": inline},
                  fh, indent=0)
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    fastq = {}
    for path in sorted(glob.glob(os.path.join(REF, "qcat", "test", "data", "*.fastq"))):
        shutil.copyfile(path, os.path.join(HERE, "data", os.path.basename(path)))
        with open(path) as fh:
            lines = fh.read().split("\n")
        fastq[os.path.basename(path)] = [(lines[i][1:], lines[i + 1]) for i in range(0, len(lines) - 3, 4)]

    cases = []

    def run_case(name, mode, kit, reads, keep_rows=False, min_quality=None, note=None, gen=None):
        det = ref_scanner.factory(mode=mode, kit=kit, min_quality=min_quality)
        layouts = det.layouts
        recs = []
        n_align0, cells0 = parasail.CALLS["n"], parasail.CALLS["cells"]
        for r in reads:
            tracer.take()
            res = det.detect_barcode(r, qcat_config=cfg)
            ends = tracer.take()
            recs.append({"result": result_to_json(res, layouts),
                         "ends": [end_to_json(e, keep_rows) for e in ends]})
        case = {"name": name, "mode": mode, "kit": kit, "min_quality": det.min_quality,
                "layout_kits": [l.kit for l in layouts],
                "layout_lens": [l.get_adapter_length() for l in layouts],
                "records": recs, "note": note}
        if gen is not None:
            case["gen"] = gen                # reads are re-generated from seeds by the tests
        else:
            case["reads"] = reads
        cases.append(case)
        print("%-34s %4d reads  %7d alignments (incl. trace re-runs)" % (name, len(reads), parasail.CALLS["n"] - n_align0))

    # 1. the 34 shipped reads, kit auto and the file's own kit
    file_kits = {"nbd103.fastq": "NBD104/NBD114", "pbk004.fastq": "PBK004/LWB001",
                 "rab204.fastq": "RAB204/RAB214", "rbk004.fastq": "RBK004"}
    for fname, recs in fastq.items():
        run_case("fastq:%s:auto" % fname, "epi2me", None, [s for _, s in recs], note=[h for h, _ in recs])
        run_case("fastq:%s:kit" % fname, "epi2me", file_kits[fname], [s for _, s in recs], keep_rows=(fname == "nbd103.fastq"))

    # 2. inline reads of the reference tests
    inl = [inline[n] for n in inline_names] + [""]
    run_case("inline:auto", "epi2me", None, inl)
    run_case("inline:RBK001", "epi2me", "RBK001", inl, keep_rows=True)
    run_case("inline:dual", "dual", None, inl, keep_rows=True)

    # 3. edge cases under PBC096
    pbc5 = [l for l in ref_scanner.get_adapter_by_name("PBC096") if l.get_adapter_length() == 59][0]
    exact = synth.fill(pbc5, 0, 0)
    body = synth.synth_read(7, 99, [pbc5], -1, -1, insert_len=400)
    edge = ["", "A", body[:32], "N" * 200, ("ACGTTGCA" + exact + body).lower(),
            body[:1], body[:30], body[:149], body[:150], body[:151], body[:299], body[:300],
            "ACGT" + exact[:30] + "NNNN" + exact[34:] + body, "RYKM" + exact + "U" * 10 + body[:200] + "*-",
            exact, exact[:40], exact[20:] + body[:100], body[:100] + synth.revcomp_acgt(exact),
            "G" * 7 + exact + body[:300] + synth.revcomp_acgt(synth.fill(pbc5, 5, 0)) + "C" * 9]
    run_case("edge:PBC096", "epi2me", "PBC096", edge, keep_rows=True)
    run_case("edge:auto", "epi2me", None, edge)
    run_case("edge:dual", "dual", None, edge)

    # 4. synthetic reads (re-generated by the tests from the stored parameters)
    def synth_case(tag, mode, kit, t5, t3, e, n, seed, keep_rows=False, bare=0.05):
        det = ref_scanner.factory(mode=mode, kit=kit)
        gen = {"seed": seed, "n": n, "tpl_5p": t5, "tpl_3p": t3, "error_rate": e,
               "no_adapter_fraction": bare, "insert_len": 600, "lead_min": 5, "lead_max": 40}
        reads = synth.synth_batch(n, seed, det.layouts, t5, t3, error_rate=e, no_adapter_fraction=bare)
        run_case("synth:%s:e%.2f" % (tag, e) + (":bare" if bare == 1.0 else ""), mode, kit, reads, keep_rows=keep_rows, gen=gen)

    seed0 = 20260928
    if r1_scalar:
        import qcat.scanner_base as sb
        assert sb.parasail_sg is parasail.sg and sb.parasail_sg is not parasail.sg_striped_32, "the reference did not bind parasail.sg"
        cases[:] = [c for c in cases if c["name"].endswith(":kit") or c["name"] in ("inline:auto", "inline:dual", "edge:PBC096")]
        synth_case("LWB001", "epi2me", "PBK004/LWB001", 1, 0, 0.08, 48, seed0 + 1)
        synth_case("NBD104", "epi2me", "NBD103/NBD104", 1, 0, 0.08, 48, seed0 + 2)
        synth_case("PBC096", "epi2me", "PBC096", 1, 0, 0.08, 48, seed0 + 3, keep_rows=True)
        synth_case("DUAL", "dual", None, 1, 0, 0.08, 48, seed0 + 5)
        synth_case("auto", "epi2me", None, 3, 2, 0.08, 32, seed0 + 7)
        synth_case("NBD104", "epi2me", "NBD103/NBD104", 1, 0, 0.08, 96, seed0 + 50, bare=1.0)
        synth_case("PBC096", "epi2me", "PBC096", 1, 0, 0.08, 96, seed0 + 51, bare=1.0)
        synth_case("RBK004", "epi2me", "RBK004", 0, -1, 0.08, 64, seed0 + 52, bare=1.0)
        with open(os.path.join(HERE, "golden_r1_scalar.json"), "w") as fh:
            json.dump({"generator": "tests/golden/make_golden.py --r1-scalar",
                       "template_order": "sorted by kit file name",
                       "r1_rule": "scalar: the stand-in's can_use_sse2() is False, the unmodified reference binds parasail.sg (qcat/scanner_base.py:20-26)",
                       "dp": "independent scalar Python DP tests/golden/sg_independent.py sg(rule='scalar')",
                       "cases": cases}, fh, separators=(",", ":"))
        print("golden_r1_scalar.json:", os.path.getsize(os.path.join(HERE, "golden_r1_scalar.json")), "bytes")
        return
    for e in (0.0, 0.08, 0.15):
        # sorted order puts the 3p template first (index 0) and the 5p template second
        synth_case("LWB001", "epi2me", "PBK004/LWB001", 1, 0, e, 48, seed0 + 1, keep_rows=(e == 0.08))
        synth_case("NBD104", "epi2me", "NBD103/NBD104", 1, 0, e, 48, seed0 + 2)
        synth_case("PBC096", "epi2me", "PBC096", 1, 0, e, 48, seed0 + 3, keep_rows=(e == 0.08))
        synth_case("DUAL", "dual", None, 1, 0, e, 48, seed0 + 5, keep_rows=(e == 0.08))
        synth_case("DUAL-epi2me", "epi2me", "DUAL", 1, 0, e, 16, seed0 + 6)
        synth_case("auto", "epi2me", None, 3, 2, e, 32, seed0 + 7)     # PBC096 reads under kit auto
    synth_case("RBK004", "epi2me", "RBK004", 0, -1, 0.08, 32, seed0 + 8)
    synth_case("VMK001", "epi2me", "VMK001", 0, -1, 0.08, 32, seed0 + 9)
    synth_case("RAB204", "epi2me", "RAB204", 1, 0, 0.08, 32, seed0 + 10)
    synth_case("RPB004", "epi2me", "RPB004/RLB001", 0, -1, 0.08, 32, seed0 + 11)

    # 5. extract_barcode_region slice table (negative-index wrap, R4)
    region_table = []
    window = body[:150]
    for kit, tidx in (("PBC096", 1), ("NBD103/NBD104", 0), ("DUAL", 1)):
        tpl = [l for l in ref_scanner.factory(mode="epi2me", kit=kit).layouts][tidx]
        for L in (150, 64, 20):
            for setidx in ((0, 1) if tpl.barcode_set_2 else (0,)):
                rows = []
                for stop in range(-1, L):
                    out = ref_base.extract_barcode_region(window[:L], tpl, setidx, stop, cfg)
                    start = window[:L].find(out) if out else 0
                    rows.append(len(out))
                region_table.append({"kit": kit, "tpl": tidx, "set": setidx, "L": L, "lens": rows})

    # 6. batch mode (detect_kit vote + detect_barcode_batch), SURVEY 8f rank 1
    batch = []
    five = [inline[n] for n in ("read", "read_bc3_exact", "read_bc3", "real_bc03_porechop", "read_nobc")]
    for kit in (None, "RBK001"):
        det = ref_scanner.factory(mode="epi2me", kit=kit)
        kit_name, _ = det.detect_kit(five, cfg)
        res = det.detect_barcode_batch(five, [None] * 5, cfg)
        batch.append({"kit": kit, "voted_kit": kit_name,
                      "results": [result_to_json(r, det.layouts) for r in res]})
    allfq = [s for recs in fastq.values() for _, s in recs]
    det = ref_scanner.factory(mode="epi2me", kit=None)
    per_file_votes = {}
    for fname, recs in fastq.items():
        kit_name, _ = det.detect_kit([s for _, s in recs], cfg)
        res = det.detect_barcode_batch([s for _, s in recs], [None] * len(recs), cfg)
        per_file_votes[fname] = {"voted_kit": kit_name,
                                 "results": [result_to_json(r, det.layouts) for r in res]}

    # 7. scan() of the 5' window only (BASELINE config 2 semantics)
    det = ref_scanner.factory(mode="epi2me", kit="NBD103/NBD104")
    reads5 = synth.synth_batch(48, seed0 + 12, det.layouts, 1, 0, error_rate=0.08)
    scan5 = []
    for r in reads5:
        w = ref_base.extract_align_sequence(r, False, cfg.max_align_length)
        tracer.take()
        res = det.scan(w, None, det.layouts, [], qcat_config=cfg)
        tracer.take()
        scan5.append(result_to_json(res, det.layouts))

    # 8. --detect-middle (scan_middle, scanner_base.py:479-519, :593-595): chimeric reads (two
    #    synthetic reads joined head to tail carry adapters in their interior), short reads, plain reads
    middle = []
    for mode, kit in (("epi2me", "PBC096"), ("epi2me", None), ("dual", None)):
        det = ref_scanner.factory(mode=mode, kit=kit, scan_middle_adapter=True)
        lays = det.layouts
        t5, t3 = (3, 2) if kit is None and mode == "epi2me" else (1, 0)
        gen = {"seed": seed0 + 20, "n": 24, "tpl_5p": t5, "tpl_3p": t3, "error_rate": 0.05}
        base = synth.synth_batch(48, gen["seed"], lays, t5, t3, error_rate=gen["error_rate"])
        reads = []
        for i in range(24):
            if i % 3 == 0:
                reads.append(base[i] + base[i + 24])                 # chimera: adapters in the middle
            elif i % 3 == 1:
                reads.append(base[i])
            else:
                reads.append(base[i][:120 + 17 * i])                 # short reads around 2 x 150
        recs = []
        for r in reads:
            tracer.take()
            res = det.detect_barcode(r, qcat_config=cfg)
            tracer.take()
            recs.append(result_to_json(res, lays))
        middle.append({"mode": mode, "kit": kit, "gen": gen, "results": recs})
        print("middle %s/%s: exit codes %s" % (mode, kit, sorted(set(x["exit_status"] for x in recs))))

    # 9. scan() of sequences longer than max_align_length and scan_middle() called directly
    #    (scanner_base.py:466-519; the form qcat/eval_full.py:199-203 uses)
    long_scan = []
    for mode, kit in (("epi2me", "PBC096"), ("epi2me", "RBK004"), ("epi2me", None), ("dual", None)):
        det = ref_scanner.factory(mode=mode, kit=kit)
        lays = det.layouts
        t5, t3 = {"PBC096": (1, 0), "RBK004": (0, -1), None: (3, 2) if mode == "epi2me" else (1, 0)}[kit]
        gen = {"seed": seed0 + 30, "n": 12, "tpl_5p": t5, "tpl_3p": t3, "error_rate": 0.08}
        base = synth.synth_batch(12, gen["seed"], lays, t5, t3, error_rate=gen["error_rate"])
        seqs = []
        for i, r in enumerate(base):
            cut = (151, 152, 200, 299, 300, 301, 450, 640, len(r), len(r), len(r), len(r))[i]
            s_ = r[:cut]
            if i >= 10:
                s_ = s_[200:] + s_[:200]          # adapter in the middle of the sequence
            seqs.append(s_)
        scans = []
        for s_ in seqs:
            tracer.take()
            res = det.scan(s_, None, lays, [], qcat_config=cfg)
            tracer.take()
            scans.append(result_to_json(res, lays))
        kit_name = lays[0].kit
        chim = [base[i] + base[i + 6] for i in range(6)] + base[:3] + [base[0][:300], base[1][:301], ""]
        middles = []
        for s_ in chim:
            tracer.take()
            middles.append(bool(det.scan_middle(s_, kit_name, cfg)))
            tracer.take()
        long_scan.append({"mode": mode, "kit": kit, "gen": gen, "scan": scans, "scan_middle_kit": kit_name,
                          "scan_middle": middles})
        print("long scan %s/%s: %d scans, scan_middle %s" % (mode, kit, len(scans), middles))

    simple = simple_section(ref_scanner, cfg, fastq, inline, inline_names, seed0)

    with open(os.path.join(HERE, "golden_vectors.json"), "w") as fh:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "template_order": "sorted by kit file name",
                   "dp": "independent scalar Python DP tests/golden/sg_independent.py (parasail absent; the oracle is not involved) -- see make_golden.py docstring",
                   "cases": cases, "region_table": region_table, "batch": batch,
                   "batch_fastq": per_file_votes, "middle": middle, "long_scan": long_scan, "simple": simple,
                   "scan5p": {"kit": "NBD103/NBD104",
                              "gen": {"seed": seed0 + 12, "n": 48, "tpl_5p": 1, "tpl_3p": 0, "error_rate": 0.08,
                                      "no_adapter_fraction": 0.05, "insert_len": 600, "lead_min": 5, "lead_max": 40},
                              "results": scan5}},
                  fh, separators=(",", ":"))
    print("golden_vectors.json:", os.path.getsize(os.path.join(HERE, "golden_vectors.json")), "bytes")


if __name__ == "__main__":
    main()
