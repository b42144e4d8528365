"""bench.py against another build of the library:  python tools/probe/bench_with_lib.py <lib.so> [bench.py arguments]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flowconductor_amd import _hip  # noqa: E402

_hip.use_library(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
