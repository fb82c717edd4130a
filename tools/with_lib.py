"""Run a tool against another build of the library: python tools/with_lib.py <lib.so> <script.py> [args...]
(for side-by-side timing of kernel variants; the tests and bench.py always load parquet-go_amd/libpqdecode.so)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parquet_go_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(sys.argv[0])))
runpy.run_path(sys.argv[0], run_name="__main__")
