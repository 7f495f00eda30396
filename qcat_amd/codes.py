"""Shared 3-bit base code space of the host layer, the C ABI and the HIP kernels.

The reference looks scores up through parasail's 256-entry mapper (alphabet letters in
either case -> index, anything else -> the ``*`` row).  Both qcat
matrices are built over ``ATGCN`` (+ ``X`` for the adapter matrix, ``qcat/config.py:26,245``),
so one code space serves both: ``A T G C N X other``.  Code 7 is used on the device only
(padding rows of ragged batches).
"""
import numpy as np

ALPHABET = "ATGCNX"
CODE_OTHER = 6
CODE_PAD = 7
NCODES = 7

ASCII_TO_CODE = np.full(256, CODE_OTHER, dtype=np.uint8)
for _i, _c in enumerate(ALPHABET):
    ASCII_TO_CODE[ord(_c)] = _i
    ASCII_TO_CODE[ord(_c.lower())] = _i


def encode(seq):
    """ASCII string/bytes -> uint8 code array."""
    if isinstance(seq, str):
        seq = seq.encode("latin-1", "replace")
    return ASCII_TO_CODE[np.frombuffer(seq, dtype=np.uint8)]
