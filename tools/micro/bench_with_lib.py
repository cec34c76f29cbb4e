"""bench.py on an experimental build of the library: python tools/micro/bench_with_lib.py LIB [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mantaflow_amd import _lib
lib = sys.argv[1]
if lib != "default":
    _lib.use_library(os.path.abspath(lib), "cuda")
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
