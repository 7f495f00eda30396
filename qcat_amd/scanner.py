"""Scanner registry: same functions as ``qcat/scanner.py`` (``factory :78-111``, ``get_modes
:33-45``, ``get_kits :63-75``, ``get_kits_info :48-60``, ``get_adapter_by_name :18-30``).  A scanner
registers itself by subclassing :class:`BarcodeScanner` and returning its mode from ``get_name()``."""
import logging

from . import adapters
from .scanner_base import BarcodeScanner
from .scanner_epi2me import BarcodeScannerEPI2ME
from .scanner_dual import BarcodeScannerDual
from .scanner_simple import BarcodeScannerSimple

__all__ = ["BarcodeScanner", "BarcodeScannerEPI2ME", "BarcodeScannerDual", "BarcodeScannerSimple", "factory",
           "get_modes", "get_kits", "get_kits_info", "get_adapter_by_name"]


def get_adapter_by_name(kit, kit_folder=None):
    return [a for a in adapters.populate_adapter_layouts(kit_folder) if a.kit == kit]


def get_modes():
    return [cls.get_name() for cls in BarcodeScanner.__subclasses__()]


def get_kits_info(kit_folder=None):
    names = {"Auto": "Auto detect kit"}
    for layout in adapters.populate_adapter_layouts(kit_folder):
        names.setdefault(layout.kit, layout.description)
    return names


def get_kits(kit_folder=None):
    names = ["Auto"]
    for layout in adapters.populate_adapter_layouts(kit_folder):
        if layout.kit not in names:
            names.append(layout.kit)
    return names


def factory(mode="epi2me", min_quality=None, kit=None, kit_folder=None,
            enable_filter_barcodes=False, scan_middle_adapter=False, threads=1, device=0):
    if mode == "guppy":
        logging.warning("Demultiplexing mode guppy is not supported; falling back to epi2me.")
        mode = "epi2me"
    for cls in BarcodeScanner.__subclasses__():
        if mode == cls.get_name():
            return cls(min_quality=min_quality, kit_folder=kit_folder, kit=kit,
                       enable_filter_barcodes=enable_filter_barcodes,
                       scan_middle_adapter=scan_middle_adapter, threads=threads, device=device)
    raise RuntimeError("Invalid demultiplexing mode: {}".format(mode))
