"""Deterministic alignment test cases for the DP pin (tests/golden/sg_vectors.json): the inputs are
re-created from (seed, index) here, the fixture stores only the expected (score, end_query,
end_ref) triples, which come from the independent scalar DP of tests/golden/sg_independent.py
(generator: tests/golden/make_sg_vectors.py).

Families (index % 8):
  0,1  adapter alignment as the reference runs it (qcat/scanner_base.py:214-218): a 150-nt window
       holding a mutated copy of a shipped template (N-run filled with a barcode) against the
       N-masked template, adapter matrix, gaps 2/2;
  2    the same against an ADAPTER-FREE random window: where the end-position ties of rule R1 live;
  3,4  barcode alignment (:111-117): an extracted region (4..150 nt) against ctx+barcode+ctx, +1/-1, gaps 1/1;
  5    random scores with open != extend (affine), random lengths;
  6    degenerate / hostile input: lengths 1..6, all-N, lower case, IUPAC and non-letters;
  7    homopolymers and short tandem repeats (maximal ties in both border scans).
"""
import numpy as np

import synth

ADAPTER_TABLE = None
BARCODE_TABLE = None
_layouts = None


def _tables():
    global ADAPTER_TABLE, BARCODE_TABLE, _layouts
    if ADAPTER_TABLE is None:
        from qcat_amd import adapters, config
        cfg = config.qcatConfig()
        ADAPTER_TABLE = np.array(cfg.matrix.table, dtype=np.int8)
        BARCODE_TABLE = np.array(cfg.matrix_barcode.table, dtype=np.int8)
        _layouts = [l for l in adapters.populate_adapter_layouts(None) if l.barcode_set_1]
    return ADAPTER_TABLE, BARCODE_TABLE, _layouts


def custom_table(match, mismatch, nmatch):
    """7x7 [target, query] table the way qcat/config.py:236-253 builds the adapter matrix."""
    t = np.zeros((7, 7), dtype=np.int8)
    for i in range(4):
        for j in range(4):
            t[i, j] = match if i == j else mismatch
    t[4, :5] = nmatch
    t[:5, 4] = nmatch
    return t


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(alphabet[rng.below(len(alphabet))] for _ in range(n))


def _mutated(rng, seq, rate):
    out = []
    synth._mutate(rng, seq, synth.rate_threshold(rate), out)
    return "".join(out)


def case(seed, index):
    """-> (s1 query, s2 target, gap_open, gap_extend, table7)."""
    ta, tb, lays = _tables()
    rng = synth.SplitMix64(seed, index)
    fam = index % 8
    if fam in (0, 1, 2):
        lay = lays[rng.below(len(lays))]
        target = lay.get_adapter_sequences()
        if fam == 2:
            window = _rand_seq(rng, 150 if rng.below(4) else 20 + rng.below(131))
        else:
            filled = synth.fill(lay, rng.below(1 << 16), rng.below(1 << 16))
            rate = (0.0, 0.05, 0.1, 0.2)[rng.below(4)]
            lead = rng.below(60)
            window = (_rand_seq(rng, lead) + _mutated(rng, filled, rate) + _rand_seq(rng, 150))[:150]
            if rng.below(8) == 0:
                window = window[:30 + rng.below(100)]
        return window, target, 2, 2, ta
    if fam in (3, 4):
        lay = lays[rng.below(len(lays))]
        s = 1 if (lay.barcode_set_2 and rng.below(2)) else 0
        bset = lay.get_barcode_set(s)
        up, down = lay.get_upstream_context(11, s), lay.get_downstream_context(11, s)
        bc = bset[rng.below(len(bset))].sequence
        target = up + bc + down
        if fam == 3:       # a noisy copy of some barcode of the set inside a short region
            other = bset[rng.below(len(bset))].sequence
            region = _rand_seq(rng, rng.below(12)) + _mutated(rng, up + other + down, (0.0, 0.08, 0.2)[rng.below(3)]) + _rand_seq(rng, rng.below(12))
            region = region[:150] or "A"
        else:              # whole-window path: unrelated 4..150 nt
            region = _rand_seq(rng, 4 + rng.below(147))
        return region, target, 1, 1, tb
    if fam == 5:
        match, mismatch, nmatch = 1 + rng.below(9), -(1 + rng.below(6)), -rng.below(3)
        extend = 1 + rng.below(4)
        gap_open = extend + rng.below(5)
        t = custom_table(match, mismatch, nmatch)
        n, m = 1 + rng.below(150), 1 + rng.below(100)
        s2 = _rand_seq(rng, m, "ACGTN" if rng.below(3) == 0 else "ACGT")
        if rng.below(2):
            s1 = (_rand_seq(rng, rng.below(40)) + _mutated(rng, s2.replace("N", "A"), 0.15) + _rand_seq(rng, 150))[:n]
        else:
            s1 = _rand_seq(rng, n)
        return s1 or "C", s2, gap_open, extend, t
    if fam == 6:
        alpha = "ACGTNacgtnRYKMSWXx*-U."
        n, m = 1 + rng.below(6 if rng.below(2) else 150), 1 + rng.below(6 if rng.below(2) else 60)
        s1 = _rand_seq(rng, n, alpha if rng.below(3) else "N")
        s2 = _rand_seq(rng, m, "ACGTNX" if rng.below(3) else "N")
        tbl, g = ((ta, 2), (tb, 1))[rng.below(2)]
        return s1, s2, g, g, tbl
    unit = _rand_seq(rng, 1 + rng.below(3))
    n, m = 1 + rng.below(150), 1 + rng.below(60)
    s1 = (unit * 150)[:n]
    s2 = ((unit if rng.below(2) else _rand_seq(rng, 1 + rng.below(3))) * 60)[:m]
    tbl, g = ((ta, 2), (tb, 1))[rng.below(2)]
    return s1, s2, g, g, tbl
