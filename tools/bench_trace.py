"""bench.py under a watchdog: dumps every thread's Python stack and exits if the run
takes longer than IBL_BENCH_WATCHDOG seconds (default 60).  For chasing hangs in
multi-rank rehearsals."""
import faulthandler
import os
import runpy
import sys

faulthandler.dump_traceback_later(int(os.environ.get("IBL_BENCH_WATCHDOG", "60")), exit=True)
sys.argv[0] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
