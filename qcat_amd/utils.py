"""Small sequence helpers of the hot path.

Reference behaviour mirrored (not copied): ``qcat/utils.py:20-21`` (``revcomp``: ACGT and
the IUPAC pairs, case preserving, everything else -- including ``N`` -- unchanged) and
``qcat/utils.py:12-17`` (``qstring_to_phred``).
"""

_PAIRS = ("AT", "CG", "RY", "MK", "VB", "HD")
_COMPLEMENT = {}
for _a, _b in _PAIRS:
    for _x, _y in ((_a, _b), (_b, _a)):
        _COMPLEMENT[ord(_x)] = _y
        _COMPLEMENT[ord(_x.lower())] = _y.lower()

#: 10**(-q/10) for q in 0..99 (same table the reference keeps, ``qcat/utils.py:6-9``)
LOOKUP = [pow(10, -.1 * q) for q in range(100)]


def revcomp(seq):
    """Reverse complement; characters without a complement are kept as they are."""
    return seq.translate(_COMPLEMENT)[::-1]


def qstring_to_phred(quality):
    """Sanger quality string -> list of phred ints (``None`` -> empty list)."""
    if quality is None:
        return []
    return [ord(c) - 33 for c in quality]


def mean_error_prob(scores):
    """Mean error probability of a list of phred scores (-1.0 for an empty list)."""
    if not scores:
        return -1.0
    return sum(LOOKUP[v] for v in scores) / len(scores)
