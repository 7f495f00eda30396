import cProfile, pstats, sys, os, io
sys.argv = ["bench_cli.py", "100000"]
pr = cProfile.Profile()
pr.enable()
exec(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "bench_cli.py")).read())
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue()[:6000])
