import cProfile, pstats, sys, io
sys.argv = ["bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--points", "1400000", "--grid", "3"]
sys.path.insert(0, "/root/repo")
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("/root/repo/bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(28)
print(s.getvalue()[:6000])
