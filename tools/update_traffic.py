#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh into profiles/<round>_traffic.json:
usage: tools/update_traffic.py <round-tag> <workload>=<gpurun_out/prof_dir>[:reads per launch, default 1000000] ..."""
import csv, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = {"_comment": "HBM traffic of the dominant kernel per launch, from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate "
       "passes, tools/profile.sh; per-dispatch averages in profiles/%s_*/summary.txt). bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: "
       "FETCH_SIZE/WRITE_SIZE are in KiB and gfx950's FETCH_SIZE reports half of a wide coalesced read stream "
       "(MI355X_MICROARCH.md, HBM section); all kernels of the dominant timing mark that ran in the step are summed (bit-sliced: k_bs_barcode of every family + k_bs_select + k_bs_plan; else the static-letter kernels of every group and the table kernels of every width class)." % tag}


def stamp(d):
    """what a replayed figure is checked against (bench.py: stale_reason): the commit the profiled build was made from (this
    tool runs in the authoring container right after the gpurun call; --commit overrides) and the per-launch duration of
    every timing mark in the un-profiled run of the same command (bench_plain.log, tools/profile.sh step 0)"""
    import subprocess
    commit = None
    for a in sys.argv:
        if a.startswith("--commit="):
            commit = a.split("=", 1)[1]
    if commit is None:
        try:
            commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
            if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "qcat_amd", "bench.py"]).decode().strip():
                commit += "+uncommitted"
        except Exception:
            commit = "unknown"
    marks = None
    try:
        with open(os.path.join(d, "bench_plain.log")) as fh:
            line = [l for l in fh.read().splitlines() if l.startswith("{")][-1]
        marks = json.loads(line)["roofline"]["kernels_avg_ms"]
    except (IOError, IndexError, KeyError, ValueError, TypeError):
        pass
    return commit, marks


for arg in [x for x in sys.argv[2:] if not x.startswith("--")]:
    wl, d = arg.split("=")
    reads = 1000000
    if ":" in d:
        d, r = d.rsplit(":", 1)
        reads = int(r)
    # the dominant mark of the step: the bit-sliced barcode kernels when they ran, else the packed-binary16 ones
    with open(os.path.join(d, "pmc3.csv")) as fh:
        bitsliced = any("k_bs_barcode" in row["Kernel_Name"] for row in csv.DictReader(fh))
    pats = ("k_bs_barcode", "k_bs_select", "k_bs_plan") if bitsliced else ("k_barcode_packed", "k_barcode_static")
    tot = defaultdict(float)
    for name, fn in (("FETCH_SIZE", "pmc3.csv"), ("WRITE_SIZE", "pmc4.csv")):
        per = defaultdict(list)
        with open(os.path.join(d, fn)) as fh:
            for row in csv.DictReader(fh):
                if any(p in row["Kernel_Name"] for p in pats) and row["Counter_Name"] == name:
                    per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        tot[name] = sum(sum(v) / len(v) for v in per.values())
    commit, marks = stamp(d)
    dom_mark = "k_barcode_bitslice" if bitsliced else "k_barcode_static"
    out[wl] = {"commit": commit, "mark_ms": (marks or {}).get(dom_mark), "kernel": "k_barcode_bitslice (k_bs_barcode + k_bs_select + k_bs_plan)" if bitsliced else "k_barcode_static+k_barcode_packed",
               "reads_per_launch": reads, "fetch_size_kib": round(tot["FETCH_SIZE"], 1),
               "write_size_kib": round(tot["WRITE_SIZE"], 1), "bytes": int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024)}
with open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
