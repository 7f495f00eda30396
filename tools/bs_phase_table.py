#!/usr/bin/env python3
"""Phase table of k_bs_barcode units from the stamps a QCAT_HIP_BS_TRACE=1 scan prints (csrc/packed_host.inc).

usage: python tools/bs_phase_table.py trace.txt [trace2.txt ...]

A line is  "[qcat] bs launch <l> unit <u> wave <w>: t0 .. t7"  (s_memtime, shader cycles relative to wave 0's t0):
  0 unit start (behind the barrier)   1 unit drawn and decoded        2 own share of the transposition done
  3 behind the barrier                4 own shared pass done          5 behind the barrier (row loops start)
  6 own barcodes done                 7 behind the unit's last barrier (the key exchange follows)
Per unit class (by row count: a unit of more than 1.2 M cycles is a full-window unit) the table gives the share of the
unit in each phase on the critical path, what the waves of a SIMD do in the row phase (when each of the four finishes)
and the wave-cycles parked at the barriers."""
import re
import sys
from collections import defaultdict

pat = re.compile(r"\[qcat\] bs launch (\d+) unit (\d+) wave\s+(\d+):\s+(.*)")


def load(path):
    units = defaultdict(dict)
    for line in open(path, errors="replace"):
        m = pat.match(line)
        if m:
            units[(path, int(m.group(1)), int(m.group(2)))][int(m.group(3))] = [int(x) for x in m.group(4).split()]
    return units


def main():
    units = {}
    for p in sys.argv[1:]:
        units.update(load(p))
    classes = defaultdict(list)
    for key, waves in units.items():
        if len(waves) != 16:
            continue
        total = max(w[7] for w in waves.values())
        classes["full window (150 rows)" if total > 1_200_000 else "nominal region (47 rows)"].append(waves)
    for name, us in sorted(classes.items()):
        n = len(us)
        acc = defaultdict(float)
        fin = [0.0] * 4
        for waves in us:
            t = lambda ev, f=max: f(w[ev] for w in waves.values())
            total = t(7)
            acc["unit cycles"] += total
            acc["draw + decode"] += t(1) / total
            acc["transposition (16 waves)"] += (t(3) - t(1)) / total
            acc["shared passes (8 of 16 waves)"] += (t(5) - t(3)) / total
            acc["row loops until the LAST wave is done"] += (t(7) - t(5)) / total
            acc["row loops until the FIRST wave is done"] += (t(6, min) - t(5)) / total
            # wave-cycles parked: at the three barriers of the prologue and at the unit's last one
            parked = sum((w[3] - w[2]) + (w[5] - w[4]) + (w[7] - w[6]) for w in waves.values())
            acc["wave-cycles parked at barriers"] += parked / (16.0 * total)
            acc["... of them behind the row loops"] += sum(w[7] - w[6] for w in waves.values()) / (16.0 * total)
            # the four waves of a SIMD (wave w sits on SIMD w % 4): when the k-th of them finishes, as a share of the row phase
            for simd in range(4):
                ends = sorted(waves[w][6] - t(5) for w in range(simd, 16, 4))
                for k in range(4):
                    fin[k] += ends[k] / (t(7) - t(5)) / 4.0
        print("%s: %d units traced, mean %.0f cycles" % (name, n, acc["unit cycles"] / n))
        for k, v in acc.items():
            if k != "unit cycles":
                print("    %-44s %5.1f %%" % (k, 100.0 * v / n))
        print("    the waves of a SIMD finish their barcodes at   %s   of the row phase" % "  ".join("%.2f" % (f / n) for f in fin))


if __name__ == "__main__":
    main()
