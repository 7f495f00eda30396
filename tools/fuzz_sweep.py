#!/usr/bin/env python3
"""Wider sweep of tests/test_hip_fuzz.py::test_random_config_shipped_kits on the GPU box:
    python tools/fuzz_sweep.py FIRST LAST      (seeds FIRST..LAST-1; the test suite itself runs 0..15)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_fuzz as fz          # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    try:
        fz.test_random_config_shipped_kits(seed)
    except AssertionError as exc:
        bad += 1
        print("seed %d FAILED: %s" % (seed, str(exc)[:200]))
print("seeds %d..%d: %d failures" % (first, last - 1, bad))
sys.exit(1 if bad else 0)
