#!/usr/bin/env python3
"""Fold the SQ_INSTS_VALU / GRBM_GUI_ACTIVE passes of tools/profile.sh into profiles/<round>_pmc.json, the file
bench.py's `valu_issue` block reads (counters instead of an instruction-mix model):

    tools/update_pmc.py <round-tag> <workload>=<prof dir>[:reads per launch, default 1000000] ...

Per workload: for every timing mark of the library (qcat_ctx_last_timing) the SQ_INSTS_VALU sum of the kernels
that run inside that mark, per scan, and the chip's effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of the
long kernels of the same run (GRBM_GUI_ACTIVE counts GPU-busy cycles per dispatch)."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel-name fragments -> the timing mark their launches are enclosed by (packed_host.inc / qcat_hip.hip)
MARKS = [
    ("k_pack_windows", "k_pack_windows"), ("k_expand_special", "k_pack_windows"), ("k_fill_multi", "k_pack_windows"),
    ("k_mid_windows", "k_middle_packed"), ("k_adapter_middle", "k_middle_packed"), ("k_middle", "k_middle_packed"), ("k_mid_", "k_middle_packed"),
    ("k_adapter_finish", "k_job_sort"),                 # (launched after the adapter phase's mark: its time is in the next one)
    ("k_abs_", "k_adapter_static"), ("k_adapter_bs", "k_adapter_static"), ("k_adapter_ms", "k_adapter_static"), ("k_adapter_mw", "k_adapter_static"),
    ("k_adapter_fused2", "k_adapter_static"), ("k_adapter_static", "k_adapter_static"), ("k_adapter_multi", "k_adapter_static"),
    ("k_adapter_packed", "k_adapter_packed"),
    ("k_job_", "k_job_sort"), ("k_bs_plan", "k_job_sort"),
    ("k_bs_", "k_barcode_bitslice"),
    ("k_barcode_static", "k_barcode_static"), ("k_barcode_multi", "k_barcode_static"), ("k_barcode_packed", "k_barcode_packed"),
    ("k_barcode_select", "k_barcode_select"), ("k_barcode_redo", "k_barcode_select"),
    ("k_finalize", "k_finalize"),
]


def mark_of(kernel):
    for frag, mark in MARKS:
        if frag in kernel:
            return mark
    return None


def per_kernel(fn, counter):
    """kernel name -> list of per-dispatch values of `counter`"""
    per = defaultdict(list)
    with open(fn) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                per[row["Kernel_Name"]].append((float(row["Counter_Value"]),
                                                float(row.get("End_Timestamp", 0) or 0) - float(row.get("Start_Timestamp", 0) or 0)))
    return per



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


def main():
    tag = sys.argv[1]
    out = {"_comment": "VALU instructions issued per scan (SQ_INSTS_VALU, summed over the kernels inside a timing mark, "
           "divided by the scans of the run) and effective clock (GRBM_GUI_ACTIVE / dispatch duration over kernels longer "
           "than 1 ms) from rocprofv3 --pmc passes of `bench.py --workload <w>` (tools/profile.sh; per-dispatch averages in "
           "profiles/%s_*/summary.txt).  bench.py: issue utilisation = insts_valu x 2 cycles / (1024 SIMDs x clock x mark "
           "time measured live)." % tag}
    for arg in [x for x in sys.argv[2:] if not x.startswith("--")]:
        wl, d = arg.split("=")
        reads = 1000000
        if ":" in d:
            d, r = d.rsplit(":", 1)
            reads = int(r)
        insts = per_kernel(os.path.join(d, "pmc1.csv"), "SQ_INSTS_VALU")
        # scans in the profiled run = dispatches of k_finalize (one per scan)
        n_scans = max([len(v) for k, v in insts.items() if "k_finalize" in k] or [1])
        marks = defaultdict(lambda: {"insts_valu": 0.0, "kernels": []})
        abs_ran = any("k_adapter_bs" in k or "k_adapter_ms" in k or "k_adapter_mw" in k for k in insts)          # the adapter phase's mark says which kernels took the batch
        for kname, vals in insts.items():
            m = mark_of(kname)
            if m is None:
                continue
            if m == "k_adapter_static" and abs_ran:
                m = "k_adapter_bitslice"
            marks[m]["insts_valu"] += sum(v for v, _ in vals) / n_scans
            short = kname.split("(")[0][:60]
            if short not in marks[m]["kernels"]:
                marks[m]["kernels"].append(short)
        clock = None
        try:
            gui = per_kernel(os.path.join(d, "pmc2.csv"), "GRBM_GUI_ACTIVE")
            cyc = ns = 0.0
            for vals in gui.values():
                for v, dt in vals:
                    if dt > 1e6:
                        cyc += v
                        ns += dt
            if ns > 0:
                clock = cyc / ns / 8.0          # rocprofv3 reports the sum over the 8 XCDs' GRBMs
                if not (0.8 <= clock <= 2.6):   # not a clock: leave the nominal figure
                    clock = None
        except IOError:
            pass
        commit, mark_ms = stamp(d)
        out[wl] = {"reads_per_launch": reads, "scans_in_run": n_scans, "commit": commit, "mark_ms": mark_ms,
                   "clock_ghz": round(clock, 4) if clock else 2.4,
                   "clock_source": "GRBM_GUI_ACTIVE (sum over 8 XCDs) / 8 / dispatch duration, kernels > 1 ms" if clock else "nominal 2.4 GHz (no GRBM pass)",
                   "marks": {m: {"insts_valu": int(v["insts_valu"]), "kernels": v["kernels"]} for m, v in sorted(marks.items())}}
    with open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
