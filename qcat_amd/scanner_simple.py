"""Simple scanner on the MI355X (mirror of ``qcat/scanner_simple.py``): no adapter templates --
every barcode of ONE list (``simple_standard`` / ``simple_extended`` of the kit bundle, or the
sequences of a FASTA file) is aligned to the whole read-end window and the best one is reported
when its score reaches ``min_quality`` (``scanner_simple.py:70-91``).

The reference calls ``find_highest_scoring_barcode(..., compute_identity=True)``
(``scanner_base.py:63-141``), which aligns with ``parasail.sg_stats*`` but returns
``(max_barcode, q_score, max_score, max_end)``: the value ``scan()`` names ``identity`` and compares
with ``min_quality`` is the normalised *score*, and the ``matches`` / ``length`` statistics never
leave that function.  So this mode needs exactly what the other modes need from the DP -- score and
``end_query`` -- and runs on the library's general int32 kernel (``QCAT_MODE_SIMPLE``).
"""
import ctypes as C
import logging
import os

import numpy as np

from . import adapters, config, native
from .adapters import get_barcodes_from_fastq, get_barcodes_simple
from .scanner_base import BarcodeScanner, build_return_dict


class _SimpleLayout(object):
    """what KitDescriptor needs from an AdapterLayout, for the one empty template of simple mode"""
    kit = "simple"
    trim_offset = 0

    def __init__(self, barcodes):
        self.barcode_set_1 = list(barcodes)
        self.barcode_set_2 = None
        self.barcode_pos_1 = self.barcode_pos_2 = adapters.NO_PLACEHOLDER

    def get_adapter_sequences(self):
        return ""

    def is_double_barcode(self):
        return False

    def get_barcode_end(self, i):
        return -1

    def get_barcode_length(self, i):
        # (the longest barcode: a FASTA may hold barcodes of unequal length, native.KitDescriptor pads the rows)
        return max(len(b.sequence) for b in self.barcode_set_1) if i == 0 and self.barcode_set_1 else 0

    def get_barcode_set(self, i):
        return self.barcode_set_1 if i == 0 else None


class BarcodeScannerSimple(BarcodeScanner):
    _native_mode = "simple"

    def __init__(self, min_quality=None, kit_folder=None, kit=None, enable_filter_barcodes=False,
                 scan_middle_adapter=False, threads=1, device=0):
        if min_quality is None:
            min_quality = 60                       # qcat/scanner_simple.py:14-15
        if threads != 1:
            logging.warning("threads is ignored: the simple scan runs on the GPU")
        super(BarcodeScannerSimple, self).__init__(min_quality, None, kit_folder=kit_folder,
                                                   enable_filter_barcodes=enable_filter_barcodes,
                                                   scan_middle_adapter=scan_middle_adapter,
                                                   device=device)
        # the reference takes a barcode FASTA path or the name of a bundled list (scanner_simple.py:27-30;
        # `kit=None` raises TypeError there as it does here)
        if os.path.isfile(kit) and os.path.exists(kit):
            self.barcodes = get_barcodes_from_fastq(kit)
        else:
            self.barcodes = get_barcodes_simple(kit)
        self._simple_layout = _SimpleLayout(self.barcodes or [])

    @staticmethod
    def get_name():
        return "simple"

    def barcode_count(self):
        return len(self.barcodes) + 1

    # -- native plumbing: one kit = the barcode list, whatever templates the caller passes ----------
    def descriptor(self, layouts=None, qcat_config=None, ends=native.ENDS_BOTH, scan_middle=None,
                   min_read_length=0, trim=False):
        if qcat_config is None:
            qcat_config = config.qcatConfig()
        if not self.barcodes:
            raise TypeError("'NoneType' object is not iterable")          # the reference iterates over None
        return native.KitDescriptor([self._simple_layout], qcat_config, mode="simple",
                                    min_quality=self.min_quality, ends=ends, scan_middle=False,
                                    min_read_length=min_read_length, trim=trim)

    def _native_kit(self, layouts, qcat_config, ends):
        return super(BarcodeScannerSimple, self)._native_kit([self._simple_layout], qcat_config, ends)

    def _records_to_dicts(self, recs, layouts):
        out = []
        den = np.maximum(recs["score_den"].astype(np.float64), 1.0)
        score = (recs["raw_score"].astype(np.float64) * 100.0 / (1.0 * den)).tolist()
        for i, b in enumerate(recs["barcode_idx"].tolist()):
            out.append(build_return_dict(self.barcodes[b] if b >= 0 else None, score[i] if b >= 0 else 0.0, None,
                                         int(recs["adapter_end"][i]), int(recs["exit_status"][i]),
                                         trim5p=int(recs["trim5p"][i]), trim3p=int(recs["trim3p"][i])))
        return out

    def _run(self, read_sequences, layouts, qcat_config, ends=native.ENDS_BOTH):
        # (the adapter layouts the base class passes are irrelevant here, like in the reference's scan())
        return super(BarcodeScannerSimple, self)._run(read_sequences, [self._simple_layout], qcat_config, ends)

    def _scan_sequences(self, sequences, layouts, qcat_config):
        return super(BarcodeScannerSimple, self)._scan_sequences(sequences, [self._simple_layout], qcat_config)

    def _batch_auto(self, read_sequences, n, qcat_config):
        return None

    def detect_kit(self, read_sequences, qcat_config=None):
        # the reference's batch mode still votes over the auto-detect adapter layouts here and then ignores
        # the outcome (scan() never looks at the templates): nothing to compute
        return None, []
