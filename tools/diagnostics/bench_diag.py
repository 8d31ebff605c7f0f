"""bench.py on the DIAGNOSTIC library (libpy4cast_hip_diag.so: the P4C_* A/B switches are live).  Same flags as bench.py; for A/B
timing of one kernel variant inside the whole step only -- the judged lines come from bench.py on the product library.
Usage: P4C_<SWITCH>=... python tools/diagnostics/bench_diag.py [bench.py flags]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from py4cast_amd import _lib  # noqa: E402

with _lib.use_diagnostic_library():
    bench.main()
