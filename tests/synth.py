"""Python twin of the stateless synthetic-read generator.

The authoritative definition is qcat_amd/csrc/synth.h (used by the device generator kernel
and by the host entry point ``qcat_synth_read``); tests check that all three agree.  Every
read is a pure function of (seed, index, parameters, templates), so shards of a batch can be
generated independently on any device.

    read = lead + M(fill(T5p)) + insert + M(revcomp(fill(T3p))) + tail

lead/tail: uniform length in [lead_min, lead_max], random ACGT; insert: ``insert_len`` random
ACGT; fill(): the template with its first N-run replaced by barcode ``b % n_set0`` and its
second N-run (double-barcode templates) by barcode ``b2 % n_set1``; M(): per-base error with
probability ``error_rate`` -- substitution / deletion / insertion with equal odds; a fraction
``no_adapter_fraction`` of the reads carries no adapter at all.
"""

MASK = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


class SplitMix64(object):
    def __init__(self, seed, index):
        self.s = (seed ^ ((index * 0xD1342543DE82EF95) & MASK)) & MASK

    def next(self):
        self.s = (self.s + GOLDEN) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def below(self, n):
        """uniform integer in [0, n) (multiply-shift on the high 32 bits)."""
        return ((self.next() >> 32) * n) >> 32

    def u24(self):
        return self.next() >> 40


def rate_threshold(rate):
    """float rate -> 24-bit integer threshold, the way the C code does it (float32 math)."""
    import numpy as np
    return int(np.float32(rate) * np.float32(16777216.0))


_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
_BASES = "ACGT"


def fill(layout, b, b2):
    seq = list(layout.get_adapter_sequences())
    for i, (pos, bset, idx) in enumerate(((layout.barcode_pos_1, layout.barcode_set_1, b),
                                          (layout.barcode_pos_2, layout.barcode_set_2, b2))):
        if bset and pos.end >= 0:
            bc = bset[idx % len(bset)].sequence
            seq[pos.start:pos.end + 1] = list(bc)
    return "".join(seq)


def revcomp_acgt(seq):
    return "".join(_COMP.get(c, c) for c in reversed(seq))


def _mutate(rng, seq, thr, out):
    for c in seq:
        if rng.u24() < thr:
            kind = rng.below(3)
            if kind == 0:                       # substitution by one of the 3 other bases
                k = _BASES.find(c)
                r = rng.below(3)
                out.append(_BASES[(k + 1 + r) & 3] if k >= 0 else _BASES[r])
            elif kind == 1:                     # deletion
                pass
            else:                               # insertion before the base
                out.append(_BASES[rng.below(4)])
                out.append(c)
        else:
            out.append(c)


def synth_read(index, seed, layouts, tpl_5p, tpl_3p, error_rate=0.0, no_adapter_fraction=0.05,
               insert_len=600, lead_min=5, lead_max=40, force_barcode=None, force_bare=None):
    """``force_barcode`` / ``force_bare`` override the drawn barcode / adapter-free flag without changing the
    random stream (config 1's file: a fixed barcode and a fixed number of adapter-free reads)."""
    rng = SplitMix64(seed, index)
    thr_err = rate_threshold(error_rate)
    thr_none = rate_threshold(no_adapter_fraction)
    bare = rng.u24() < thr_none
    span = lead_max - lead_min + 1
    lead = lead_min + rng.below(span)
    tail = lead_min + rng.below(span)
    b = rng.below(1 << 16)
    b2 = rng.below(1 << 16)
    if force_barcode is not None:
        b = force_barcode
    if force_bare is not None:
        bare = force_bare
    out = []
    for _ in range(lead):
        out.append(_BASES[rng.below(4)])
    if not bare and tpl_5p >= 0:
        _mutate(rng, fill(layouts[tpl_5p], b, b2), thr_err, out)
    for _ in range(insert_len):
        out.append(_BASES[rng.below(4)])
    if not bare and tpl_3p >= 0:
        _mutate(rng, revcomp_acgt(fill(layouts[tpl_3p], b, b2)), thr_err, out)
    for _ in range(tail):
        out.append(_BASES[rng.below(4)])
    return "".join(out)


def synth_batch(n, seed, layouts, tpl_5p, tpl_3p, first=0, **kw):
    return [synth_read(first + i, seed, layouts, tpl_5p, tpl_3p, **kw) for i in range(n)]


CONFIG1 = {"seed": 20260928, "n_barcoded": 191, "n_bare": 2, "error_rate": 0.08, "insert_len": 300,
           "kit": "PBK004/LWB001", "bare_at": (57, 140)}


def config1_fastq(layouts):
    """BASELINE config 1 (ii): the README's 193-read example as a generated FASTQ -- 191 reads of kit
    PBK004/LWB001 carrying ``barcode01`` at both ends (8 % errors) and 2 adapter-free reads.  ``layouts`` = the kit's
    templates in sorted order (3p, 5p).  Returns the file's text; names read000..read192, constant qualities."""
    c = CONFIG1
    lines = []
    for i in range(c["n_barcoded"] + c["n_bare"]):
        seq = synth_read(i, c["seed"], layouts, 1, 0, error_rate=c["error_rate"], insert_len=c["insert_len"],
                         force_barcode=0, force_bare=(i in c["bare_at"]))
        lines += ["@read%03d runid=config1 ch=%d" % (i, 1 + i % 512), seq, "+", "I" * len(seq)]
    return "\n".join(lines) + "\n"


FLAGS = {"seed": 20260930, "n": 150, "error_rate": 0.06, "insert_len": 120, "kit": "NBD104/NBD114"}


def flags_fastq(layouts):
    """A small file on which the driver's rarely used flags DO something (round 5: --detect-middle and --filter-barcodes on
    the native file path; golden runs of the reference driver with the flags set, tests/golden/make_cli_golden.py): 150 reads of
    kit NBD104/NBD114 -- 60 % barcode 1, 30 % barcode 2, every tenth read one of a dozen barcodes seen once or twice (the
    per-batch filter of qcat/scanner_base.py:690-712 drops those), every sixth read a chimera of a read with itself or with its
    reverse complement (an adapter with the same barcode in the interior: exit status 997 under --detect-middle,
    scanner_base.py:479-519, :593-595).  ``layouts`` = the kit's templates in sorted order.  Returns the file's text."""
    c = FLAGS
    lines = []
    for i in range(c["n"]):
        b = (2 + i // 10) if i % 10 == 9 else (0 if i % 10 < 6 else 1)
        seq = synth_read(i, c["seed"], layouts, 1, 0, error_rate=c["error_rate"], insert_len=c["insert_len"], force_barcode=b)
        if i % 6 == 2:
            seq = seq + (revcomp_acgt(seq) if i % 12 == 2 else seq)
        lines += ["@read%03d runid=flags ch=%d" % (i, 1 + i % 512), seq, "+", "I" * len(seq)]
    return "\n".join(lines) + "\n"
