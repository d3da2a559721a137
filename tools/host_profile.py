"""Host-side cost of a training step (the B=32, S=128 configuration is launch-bound): cProfile of 30 steps."""
import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--batch", "32", "--seq", "128", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
