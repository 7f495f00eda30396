"""Kit data model and loaders: host-side mirror of ``qcat.adapters``.

``Barcode`` has the reference's fields (``qcat/adapters.py:15``).  ``populate_adapter_layouts``
keeps the reference's contract (``qcat/adapters.py:138-162``: folder of ``*.yml`` kit files, a
single file, or the built-in kits) with one documented difference: layouts are returned in
**sorted file-name order** (the reference uses unsorted ``glob`` order, which makes exact
template ties filesystem dependent -- SURVEY.md section 8a, R8).

The built-in kits ship as one JSON bundle (``resources/kits.json``, generated from the kit
data files by ``tools/import_kits.py``); a user ``kit_folder`` is parsed from YAML files in
the reference's format (``qcat/adapters.py:75-105``).
"""
import glob
import json
import logging
import os
from collections import namedtuple

from .layout import AdapterLayout

Barcode = namedtuple("Barcode", "name id sequence fwd_strand")
_Placeholder = namedtuple("_Placeholder", "start end")
NO_PLACEHOLDER = _Placeholder(-1, -1)      # what AdapterLayout keeps for a barcode set that does not exist

KIT_BUNDLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources", "kits.json")
KIT_FOLDER = KIT_BUNDLE      # name kept for callers that print it


def read_barcode(data):
    if not data:
        return None
    return Barcode(data["name"], data["id"], data.get("sequence", None),
                   data.get("fwd_strand", None))


def read_barcode_set(data):
    if not data:
        return None
    return [read_barcode(item) for item in data]


def barcode2yaml(bc):
    return {"name": bc.name, "id": bc.id, "sequence": bc.sequence, "fwd_strand": bc.fwd_strand}


def barcodes2yaml(barcodes):
    return [barcode2yaml(bc) for bc in barcodes or []]


def adapter2yaml(adapter):
    return {"kit": adapter.kit,
            "auto_detect": adapter.auto_detect,
            "model": adapter.model,
            "description": adapter.description,
            "sequence": adapter.get_adapter_sequences(),
            "barcode_set_1": barcodes2yaml(adapter.get_barcode_set(0)),
            "barcode_set_2": barcodes2yaml(adapter.get_barcode_set(1))}


def _layout_from_dict(data):
    """dict in the kit-file schema -> AdapterLayout, or None for ``active: false``."""
    if not data.get("active", True):
        return None
    model = data.get("model") or None
    return AdapterLayout(kit=data.get("kit", ""),
                         sequence=data.get("sequence", ""),
                         barcode_set_1=read_barcode_set(data.get("barcode_set_1", None)),
                         barcode_set_2=read_barcode_set(data.get("barcode_set_2", None)),
                         description=data.get("description", ""),
                         auto_detect=data.get("auto_detect", False),
                         trim_offset=data.get("trim_offset", 0),
                         model=model.get("file", None) if model else None,
                         model_len=model.get("length", None) if model else None)


def read_adapter_layout(filename):
    """Parse one kit YAML file (reference format)."""
    import yaml
    with open(filename, "r") as stream:
        return _layout_from_dict(yaml.safe_load(stream))


def _bundle_entries():
    with open(KIT_BUNDLE) as fh:
        bundle = json.load(fh)
    for entry in bundle["layouts"]:
        data = dict(entry)
        for key in ("barcode_set_1", "barcode_set_2"):
            if data[key] is not None:
                data[key] = [dict(zip(Barcode._fields, row)) for row in data[key]]
        yield data


def get_barcodes_from_fastq(reads_fa):
    """Barcode list from a FASTA file, ids 1.. in file order (``qcat/adapters.py:108-118``)."""
    # the same reading rules as SimpleFastaParser, which the reference uses here: interior blanks and
    # carriage returns of a sequence are dropped (one helper for the driver and this loader)
    from .cli import _fasta_records
    barcodes = []
    with open(reads_fa) as fh:
        for title, seq in _fasta_records(fh):
            barcodes.append(read_barcode({"name": title, "id": len(barcodes) + 1, "sequence": seq}))
    if len(barcodes) <= 0:
        logging.error("Couldn't find barcodes in {}".format(reads_fa))
    return barcodes


def get_barcodes_simple(kit="standard", filename=None):
    """Barcode list of the inactive ``simple_<kit>`` entries (``qcat/adapters.py:121-135``)."""
    if filename and os.path.isfile(filename):
        import yaml
        with open(filename) as fh:
            return read_barcode_set(yaml.safe_load(fh).get("barcode_set_1", []))
    wanted = "simple_{}.yml".format(kit)
    for data in _bundle_entries():
        if data["file"] == wanted:
            return read_barcode_set(data.get("barcode_set_1", []))
    raise IOError("no such simple barcode list: {}".format(kit))


def populate_adapter_layouts(folder=None):
    """All active adapter layouts, in sorted file-name order."""
    if folder:
        if os.path.exists(folder):
            if os.path.isdir(folder):
                filenames = sorted(glob.glob(os.path.join(folder, "*.yml")))
            else:
                filenames = [folder]
            layouts = [read_adapter_layout(f) for f in filenames]
            return [l for l in layouts if l]
        logging.warning("{} not found. Using default adapter sequences.".format(folder))
    return [l for l in (_layout_from_dict(d) for d in _bundle_entries()) if l]
