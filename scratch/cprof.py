import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
