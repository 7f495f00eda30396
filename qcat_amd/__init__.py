"""qcat_amd -- MI355X-native barcode-demultiplexing hot path behind qcat's scanner API.

See DESIGN.md.  The package holds the host-side mirror of the reference's scanner interface
(`scanner.factory`, `BarcodeScannerEPI2ME`, `BarcodeScannerDual`, `qcatConfig`, kit loader) and
`csrc/` (the HIP kernels and the C-ABI shared library the scanners call through ctypes).
"""
__version__ = "0.1.0"
