"""cProfile of the config-5 run (tools/config5_run.py seed=7): where the HOST spends the run's wall time.
    python tools/config5_hostprof.py > gpurun_out/r06_config5_hostprof.txt"""
import cProfile
import io
import os
import pstats
import runpy
import sys

sys.argv = ["config5_run.py", "seed=7"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config5_run.py"), run_name="__main__")
finally:
    pr.disable()
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print(s.getvalue()[:9000])
