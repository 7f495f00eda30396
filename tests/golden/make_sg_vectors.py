#!/usr/bin/env python3
"""Generate tests/golden/sg_vectors.json: (score, end_query, end_ref) of N deterministic alignment
cases (tests/sg_cases.py) computed by the INDEPENDENT scalar Python DP (sg_independent.py) -- never
by the oracle.  tests/test_oracle_golden.py::test_dp_against_independent_vectors replays the cases
through oracle/qcat_oracle.c's qo_sg and compares.  Dev tool (a few minutes on 8 cores)."""
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import sg_cases  # noqa: E402
import sg_independent  # noqa: E402

SEED = 20260928
N = 12000


def one(index):
    s1, s2, go, ge, table = sg_cases.case(SEED, index)
    score, eq, er = sg_independent.sg(s1, s2, go, ge, sg_independent.scorer_from_table7(table.tolist()))
    # is this an end-position tie between the two border scans?  (diagnostic only)
    return [score, eq, er, len(s1), len(s2)]


def main():
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        rows = pool.map(one, range(N), chunksize=64)
    col_end = sum(1 for r in rows if r[2] == r[4] - 1)
    row_end = sum(1 for r in rows if r[1] == r[3] - 1)
    both = sum(1 for r in rows if r[2] == r[4] - 1 and r[1] == r[3] - 1)
    out = {"generator": "tests/golden/make_sg_vectors.py", "dp": "tests/golden/sg_independent.py (scalar Python, independent of the oracle)",
           "seed": SEED, "n": N,
           "stats": {"ends_in_last_column": col_end, "ends_in_last_row": row_end, "ends_in_corner": both,
                     "cells": sum(r[3] * r[4] for r in rows)},
           "results": [r[:3] for r in rows]}
    path = os.path.join(HERE, "sg_vectors.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes", out["stats"])


if __name__ == "__main__":
    main()
