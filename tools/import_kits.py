#!/usr/bin/env python3
"""Dev tool (this container only): read the kit definition data files shipped with
the reference (``/root/reference/qcat/resources/kits/*.yml``) and re-emit the *data*
(adapter template strings, barcode tables, trim offsets, flags) as one compact JSON
bundle, ``qcat_amd/resources/kits.json``.

The bundle is ordered by source file name (rule R8: the reference
uses unsorted ``glob`` order, ``qcat/adapters.py:144``; this build fixes sorted order).
Inactive entries (``active: false``, ``qcat/adapters.py:85-86``) are kept with
``"active": false`` so ``get_barcodes_simple``-style lookups stay possible.

Usage: python tools/import_kits.py [/root/reference/qcat/resources/kits]
"""
import glob
import json
import os
import sys

import yaml


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/qcat/resources/kits"
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "qcat_amd", "resources", "kits.json")
    entries = []
    for path in sorted(glob.glob(os.path.join(src, "*.yml"))):
        with open(path) as fh:
            d = yaml.safe_load(fh)

        def bset(key):
            rows = d.get(key) or None
            if not rows:
                return None
            return [[r["name"], r["id"], r.get("sequence"), r.get("fwd_strand")] for r in rows]

        model = d.get("model") or None
        entries.append({
            "file": os.path.basename(path),
            "active": bool(d.get("active", True)),
            "kit": d.get("kit", ""),
            "name": d.get("name"),
            "description": d.get("description", ""),
            "auto_detect": bool(d.get("auto_detect", False)),
            "trim_offset": int(d.get("trim_offset", 0)),
            "sequence": d.get("sequence", ""),
            "model": model,
            "barcode_set_1": bset("barcode_set_1"),
            "barcode_set_2": bset("barcode_set_2"),
        })
    with open(out, "w") as fh:
        json.dump({"format": 1, "order": "sorted-by-file-name", "layouts": entries}, fh,
                  separators=(",", ":"))
        fh.write("\n")
    print("wrote", out, len(entries), "entries", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
