#!/usr/bin/env python3
"""Generate tests/golden/simple_ragged.json: the UNMODIFIED reference's BarcodeScannerSimple (qcat/scanner_simple.py:41-91)
over a barcode FASTA whose barcodes have UNEQUAL lengths -- the reference aligns every barcode with its own length and
normalises by it (scanner_base.py:108-119).  Dev tool for the authoring container (imports /root/reference through the
stand-ins of make_golden.py: parasail -> tests/golden/sg_independent.py, Bio -> two parsers); nothing in tests/ calls it.

The FASTA: the first twelve barcodes of the bundled `standard` list, cut to 16 / 20 letters or lengthened by a few
letters in a fixed pattern.  Reads: the seeded generator's LWB001 reads (they carry those barcodes) + degenerate ones."""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import synth  # noqa: E402  (tests/synth.py, on the path through make_golden)


def ragged_fasta_text(barcodes):
    out = []
    tail = "ACGTTGCA"
    for i, b in enumerate(barcodes[:12]):
        seq = b.sequence
        k = i % 4
        if k == 0:
            seq = seq[:16]
        elif k == 1:
            seq = seq[:20]
        elif k == 3:
            seq = seq + tail[:2 + i % 5]
        out.append(">rb%02d\n%s\n" % (i + 1, seq))
    return "".join(out)


def main():
    mg.install_standins()
    ref_scanner, _b, _e, _d, ref_config = mg.import_reference()
    cfg = ref_config.qcatConfig()
    std = ref_scanner.factory(mode="simple", kit="standard").barcodes
    text = ragged_fasta_text(std)
    tmp = tempfile.mkdtemp()
    fa = os.path.join(tmp, "ragged.fasta")
    with open(fa, "w") as fh:
        fh.write(text)
    det = ref_scanner.factory(mode="simple", kit=fa)
    lens = [len(b.sequence) for b in det.barcodes]
    assert len(set(lens)) > 1
    lwb = ref_scanner.factory(mode="epi2me", kit="PBK004/LWB001").layouts
    out = []
    for e in (0.0, 0.1):
        gen = {"seed": 20260928 + 77, "n": 60, "tpl_5p": 1, "tpl_3p": 0, "error_rate": e, "kit": lwb[0].kit}
        reads = synth.synth_batch(gen["n"], gen["seed"], lwb, 1, 0, error_rate=e)
        extra = ["", "A", reads[0][:100], reads[1][:151], "N" * 200, reads[2].lower(), std[0].sequence[:16], std[3].sequence + "ACG"]
        res = []
        for r in reads + extra:
            d = det.detect_barcode(r, qcat_config=cfg)
            bc = d["barcode"]
            res.append({"barcode_index": -1 if bc is None else det.barcodes.index(bc),
                        "barcode_name": None if bc is None else bc.name, "barcode_id": None if bc is None else bc.id,
                        "score_hex": float(d["barcode_score"]).hex(), "adapter": d["adapter"],
                        "adapter_end": d["adapter_end"], "trim5p": d["trim5p"], "trim3p": d["trim3p"],
                        "exit_status": d["exit_status"]})
        out.append({"fasta": text, "lengths": lens, "min_quality": det.min_quality, "gen": gen, "extra": extra, "results": res})
        print("simple/ragged e=%.2f: %d reads, %d called, winners of length %s" % (
            e, len(res), sum(1 for x in res if x["barcode_name"]),
            sorted(set(lens[x["barcode_index"]] for x in res if x["barcode_index"] >= 0))))
    with open(os.path.join(HERE, "simple_ragged.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("simple_ragged.json:", os.path.getsize(os.path.join(HERE, "simple_ragged.json")), "bytes")


if __name__ == "__main__":
    main()
