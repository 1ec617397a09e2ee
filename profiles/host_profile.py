import cProfile, pstats, sys, io
sys_points = sys.argv[1] if len(sys.argv) > 1 else "1400000"
sys.argv = ["bench.py", "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--points", sys_points, "--grid", "3"]
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
ps.print_stats(45)
print(s.getvalue()[:9000])
for fn in ("_cuda_getDeviceCount", "is_available", "_get_available_device_type", "_get_device_attr"):
    s2 = io.StringIO()
    pstats.Stats(pr, stream=s2).sort_stats("tottime").print_callers(fn)
    print(s2.getvalue()[:3500])
