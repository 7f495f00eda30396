#!/usr/bin/env python3
"""Where a `qcat_amd.cli --tsv` run of a synthetic FASTQ file spends its wall time (cProfile, cumulative): dev tool."""
import cProfile, io, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "1000000", "2000"]
exec(open(os.path.join(ROOT, "tools", "bench_cli.py")).read().split("res = {")[0])     # helpers + the file writer
big = os.path.join(tmp, "big.fastq"); write_fastq(big, n)
native.FastqFile(big).close()
for rep in range(2):
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    run(big, "PBC096", None, True, True)
    pr.disable(); dt = time.perf_counter() - t0
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
    print("run %d: %.3f s for %d reads" % (rep, dt, n)); print("\n".join(s.getvalue().splitlines()[4:32]))
