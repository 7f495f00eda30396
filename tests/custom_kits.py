"""Custom kit folders (YAML, the reference's kit format: qcat/adapters.py:53-105) written on the
fly for the tests and for bench.py's `dual96` workload (BASELINE config 5: a dual kit whose two
barcode sets both hold 96 entries -- no such kit ships with qcat)."""
import os
import random
import tempfile

import yaml


def write_kit(folder, name, kit, seq, set1, set2=None, trim_offset=0):
    def rows(bcs):
        return [{"name": "barcode%02d" % (i + 1), "id": i + 1, "sequence": s, "fwd_strand": True} for i, s in enumerate(bcs)]
    data = {"kit": kit, "auto_detect": False, "description": "test kit", "sequence": seq, "trim_offset": trim_offset,
            "barcode_set_1": rows(set1), "barcode_set_2": rows(set2) if set2 else []}
    with open(os.path.join(folder, name + ".yml"), "w") as fh:
        yaml.safe_dump(data, fh)


def random_barcodes(rng, n, length=24, alphabet="ACGT"):
    return ["".join(rng.choice(alphabet) for _ in range(length)) for _ in range(n)]


def write_dual_96x96(folder, seed=96):
    """The DUAL kit's two templates with 96 random 24-nt barcodes in each set."""
    rng = random.Random(seed)
    s1, s2 = random_barcodes(rng, 96), random_barcodes(rng, 96)
    write_kit(folder, "DUAL_3p", "DUAL", "GGTTAA" + "N" * 24 + "CAGCACCTGGTGCTG" + "N" * 24 + "TTAACCTACTTGCC", s1, s2)
    write_kit(folder, "DUAL_5p", "DUAL", "AGGTTAA" + "N" * 24 + "CAGCACCTGGTGCTG" + "N" * 24 + "TTAACCTTTCTGTTGGTGCTGATATTGC", s1, s2)
    return folder


_tmp = []


def dual_96x96_folder(seed=96):
    """A fresh temporary kit folder holding the 96 x 96 dual kit (kept alive for the process)."""
    d = tempfile.TemporaryDirectory(prefix="qcat_dual96_")
    _tmp.append(d)
    return write_dual_96x96(d.name, seed)
