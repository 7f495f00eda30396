"""Adapter template geometry: host-side mirror of ``qcat.layout.AdapterLayout``.

Same accessor names and results as ``qcat/layout.py`` (``get_placeholder_pos :72-96``,
``get_barcode_end/length :102-130``, ``get_adapter_sequences :132-145``, contexts
``:191-238``, ``is_double_barcode :240-248``), written around one pre-computed list of
N-run placeholders.  ``pack()`` flattens the geometry for the native kit descriptor.
"""
import re
from collections import namedtuple

BarcodePosition = namedtuple("BarcodePosition", "start end length")

_NO_POS = BarcodePosition(-1, -1, 0)
_N_RUN = re.compile("N+")
_BAD_CHAR = re.compile("[^ATGCNX]")


class AdapterLayout(object):

    def __init__(self, kit, sequence, barcode_set_1, barcode_set_2, description,
                 auto_detect=False, model=None, model_len=None, name=None, trim_offset=0):
        self.kit = kit
        self.name = name if name else kit
        self.description = description
        self.auto_detect = auto_detect
        self.model = model
        self.model_len = model_len
        self.trim_offset = trim_offset

        self.sequence = sequence.upper()
        if not self.sequence or _BAD_CHAR.search(self.sequence):
            raise RuntimeError("Invalid adapter sequence: {}".format(self.sequence))

        self.barcode_set_1 = barcode_set_1
        self.barcode_set_2 = barcode_set_2
        self.barcode_count = sum(1 for s in (barcode_set_1, barcode_set_2) if s)

        self.barcode_pos_1 = self._checked_pos(barcode_set_1, 0)
        self.barcode_pos_2 = self._checked_pos(barcode_set_2, 1)

    def _checked_pos(self, barcode_set, index):
        if not barcode_set:
            return _NO_POS
        pos = self.get_placeholder_pos(self.sequence, index)
        for barcode in barcode_set:
            if len(barcode.sequence) != pos.length:
                raise RuntimeError("Adapter length does not match place holder length: "
                                   "{}, {}".format(len(barcode.sequence), pos.length))
        return pos

    @staticmethod
    def get_placeholder_pos(adapter_template, index=0):
        """(start, end, length) of the ``index``-th run of N in ``adapter_template``."""
        runs = [m.span() for m in _N_RUN.finditer(adapter_template)]
        if index < len(runs):
            lo, hi = runs[index]
            return BarcodePosition(lo, hi - 1, hi - lo)
        return _NO_POS

    def __repr__(self):
        return repr({"Kit": self.kit, "Description": self.description})

    def _pos(self, index):
        if index == 0:
            return self.barcode_pos_1
        if index == 1:
            return self.barcode_pos_2
        raise RuntimeError("Invalid barcode index: {}. Must be 0 or 1 "
                           "(for double barcoding)".format(index))

    def get_barcode_end(self, index=0):
        return self._pos(index).end

    def get_barcode_length(self, index=0):
        return self._pos(index).length

    def get_adapter_sequences(self, barcode_seq=None):
        if not barcode_seq:
            return self.sequence
        p = self.barcode_pos_1
        return self.sequence[:p.start] + barcode_seq + self.sequence[p.end + 1:]

    def get_full_adapter_sequences(self, context=None):
        if self.barcode_count == 0:
            yield None, self.sequence
        elif self.barcode_count == 1:
            p = self.barcode_pos_1
            head, tail = self.sequence[:p.start], self.sequence[p.end + 1:]
            for barcode in self.barcode_set_1:
                if context:
                    yield barcode, head[:-context] + barcode.sequence + tail[context:]
                else:
                    yield barcode, head + barcode.sequence + tail

    def get_adapter_length(self):
        return len(self.sequence)

    def get_barcode_set(self, index=0):
        self._pos(index)          # same index validation as the accessors above
        return self.barcode_set_1 if index == 0 else self.barcode_set_2

    def get_upstream_context(self, n, index=0):
        p = self._pos(index)
        if p.end < 0:
            return ""
        return self.sequence[max(0, p.start - n):p.start]

    def get_downstream_context(self, n, index=0):
        p = self._pos(index)
        if p.end < 0:
            return ""
        return self.sequence[p.end + 1:min(len(self.sequence), p.end + n + 1)]

    def is_double_barcode(self):
        return self.barcode_set_2 is not None
