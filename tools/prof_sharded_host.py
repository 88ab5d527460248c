import cProfile, pstats, sys, os
sys.argv = ["bench.py", "--dist-driver", "--n", "128", "--no-cpu-baseline", "--no-pmc", "--no-csr-section", "--steps", "8", "--warmup", "2"]
sys.path.insert(0, os.getcwd())
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
st = pstats.Stats(pr, stream=sys.stderr)
st.sort_stats("tottime").print_stats(28)
