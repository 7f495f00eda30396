"""Dual/combinatorial barcode scanner on the MI355X (mirror of ``qcat/scanner_dual.py``)."""
import logging

from .scanner_base import BarcodeScanner


class BarcodeScannerDual(BarcodeScanner):
    _native_mode = "dual"

    def __init__(self, min_quality=None, kit_folder=None, kit=None, enable_filter_barcodes=False,
                 scan_middle_adapter=False, threads=1, device=0):
        if min_quality is None:
            min_quality = 60                       # qcat/scanner_dual.py:15-16
        if threads != 1:
            logging.warning("threads is ignored: the dual scan runs on the GPU")
        # the reference forces the kit named "dual" whatever `kit` says (scanner_dual.py:23-24)
        super(BarcodeScannerDual, self).__init__(min_quality, "dual", kit_folder=kit_folder,
                                                 enable_filter_barcodes=enable_filter_barcodes,
                                                 scan_middle_adapter=scan_middle_adapter,
                                                 device=device)
        self.barcodes = None

    @staticmethod
    def get_name():
        return "dual"
