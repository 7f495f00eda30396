"""EPI2ME scanner on the MI355X (mirror of ``qcat/scanner_epi2me.py``)."""
import logging

from .scanner_base import BarcodeScanner


class BarcodeScannerEPI2ME(BarcodeScanner):
    _native_mode = "epi2me"

    def __init__(self, min_quality=None, kit_folder=None, kit=None, enable_filter_barcodes=False,
                 scan_middle_adapter=False, threads=1, device=0):
        if min_quality is None:
            min_quality = 58                       # qcat/scanner_epi2me.py:13-14
        if threads != 1:
            logging.warning("threads is ignored: the epi2me scan runs on the GPU")
        super(BarcodeScannerEPI2ME, self).__init__(min_quality, kit, kit_folder=kit_folder,
                                                   enable_filter_barcodes=enable_filter_barcodes,
                                                   scan_middle_adapter=scan_middle_adapter,
                                                   device=device)
        self.barcodes = None

    @staticmethod
    def get_name():
        return "epi2me"
