"""Scoring configuration of the hot path: a host-side mirror of ``qcat.config.qcatConfig``.

Field names, defaults and the setter quirks follow ``qcat/config.py:10-26`` (defaults),
``:42-144`` (setters: ``match``/``nmatch``/``gap_*`` store ``abs(value)``, ``mismatch`` stores
``-abs(value)``) and ``:236-253`` (adapter matrix: ACGT match/mismatch, anything against
``N`` = ``nmatch``, anything against ``X`` or an unknown character = 0).  The barcode matrix
is the fixed ``ATGCN`` +1/-1 matrix of ``qcat/config.py:26``.

The matrices are plain ``numpy.int8`` 7x7 tables over :mod:`qcat_amd.codes` (no parasail);
they are what gets uploaded into the kit descriptor of the C ABI (``include/qcat_hip.h``).
"""
import configparser

import numpy as np

from .codes import NCODES

_INI_SECTION = "qcat"
_INI_INT_FIELDS = ("gap_open", "gap_extend", "match", "mismatch", "max_align_length",
                   "extracted_barcode_extension", "barcode_context_length")


class ScoreMatrix(object):
    """7x7 substitution table ``table[target_code, query_code]`` (symmetric for qcat)."""

    def __init__(self, table):
        self.table = np.ascontiguousarray(table, dtype=np.int8).reshape(NCODES, NCODES)

    def score(self, a, b):
        from .codes import ASCII_TO_CODE
        return int(self.table[ASCII_TO_CODE[ord(b)], ASCII_TO_CODE[ord(a)]])

    def __eq__(self, other):
        return isinstance(other, ScoreMatrix) and np.array_equal(self.table, other.table)


def _adapter_table(match, mismatch, nmatch):
    t = np.zeros((NCODES, NCODES), dtype=np.int64)
    t[:4, :4] = mismatch
    for i in range(4):
        t[i, i] = match
    t[4, :5] = nmatch          # N row against A,T,G,C,N
    t[:5, 4] = nmatch          # N column
    # X (code 5) and "other" (code 6) rows/columns stay 0
    return t


def _barcode_table():
    t = np.zeros((NCODES, NCODES), dtype=np.int64)
    t[:5, :5] = -1
    for i in range(5):
        t[i, i] = 1            # N-N counts as a match in the barcode matrix
    # X is not in the barcode alphabet -> behaves like "other" (0)
    return t


class qcatConfig(object):

    def __init__(self, config_path=None):
        self._match = 5
        self._nmatch = -1
        self._mismatch = -2
        self._gap_open = 2
        self._gap_extend = 2
        self._max_align_length = 150
        self._extracted_barcode_extension = 11
        self._barcode_context_length = 11
        self.matrix = None
        self.update_matrix()
        self._matrix_barcode = ScoreMatrix(_barcode_table())
        if config_path is not None:
            self.read(config_path)

    # -- scoring values ---------------------------------------------------------------
    @property
    def matrix_barcode(self):
        return self._matrix_barcode

    @property
    def match(self):
        return self._match

    @match.setter
    def match(self, value):
        self._match = abs(value)
        self.update_matrix()

    @property
    def nmatch(self):
        return self._nmatch

    @nmatch.setter
    def nmatch(self, value):
        self._nmatch = abs(value)       # sic: the reference flips the sign here
        self.update_matrix()

    @property
    def mismatch(self):
        return self._mismatch

    @mismatch.setter
    def mismatch(self, value):
        self._mismatch = -1 * abs(value)
        self.update_matrix()

    @property
    def gap_open(self):
        return self._gap_open

    @gap_open.setter
    def gap_open(self, value):
        self._gap_open = abs(value)

    @property
    def gap_extend(self):
        return self._gap_extend

    @gap_extend.setter
    def gap_extend(self, value):
        self._gap_extend = abs(value)

    # -- geometry -----------------------------------------------------------------------
    @property
    def max_align_length(self):
        return self._max_align_length

    @max_align_length.setter
    def max_align_length(self, value):
        self._max_align_length = value

    @property
    def extracted_barcode_extension(self):
        return self._extracted_barcode_extension

    @extracted_barcode_extension.setter
    def extracted_barcode_extension(self, value):
        self._extracted_barcode_extension = value

    @property
    def barcode_context_length(self):
        return self._barcode_context_length

    @barcode_context_length.setter
    def barcode_context_length(self, value):
        self._barcode_context_length = value

    # -- matrix / ini -------------------------------------------------------------------
    def update_matrix(self):
        self.matrix = ScoreMatrix(_adapter_table(self._match, self._mismatch, self._nmatch))

    def fingerprint(self):
        """Hashable summary of everything the native kit descriptor depends on."""
        return (self._match, self._nmatch, self._mismatch, self._gap_open, self._gap_extend,
                self._max_align_length, self._extracted_barcode_extension,
                self._barcode_context_length)

    def write(self, out_config_path):
        ini = configparser.RawConfigParser()
        ini.add_section(_INI_SECTION)
        for key in _INI_INT_FIELDS:
            ini.set(_INI_SECTION, key, str(getattr(self, key)))
        with open(out_config_path, "w") as fh:     # the reference opens 'wb' (a py2-ism)
            ini.write(fh)

    def read(self, config_path):
        ini = configparser.RawConfigParser()
        ini.read(config_path)
        for key in _INI_INT_FIELDS:
            setattr(self, key, ini.getint(_INI_SECTION, key))


def get_default_config():
    return qcatConfig()
