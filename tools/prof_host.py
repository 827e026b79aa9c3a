import cProfile, pstats, sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--steps", "200", "--warmup", "8", "--no-cpu", "--no-eval"]
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
