#!/usr/bin/env python3
"""Same-box A/B of two builds of the library: load the given .so first, then run bench.py's main() with the remaining arguments
(bench.py's own load_library() call then returns the library already loaded).  Usage: ab_bench.py <lib.so> [bench.py flags]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
from sgmse_amd import _lib
_lib.load_library(lib)
import bench
bench.main()
