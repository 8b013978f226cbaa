#!/usr/bin/env python3
"""Experiments only: run bench.py against ANOTHER build of the library (e.g. one compiled with an experimental -D macro):
    python devtools/ab_lib.py vdetlib_amd/libvdet_hip_x1.so --profile --no-sharded-leg --streams 1 --steps 6
Nothing in the product reads this; the product always loads vdetlib_amd/libvdet_hip.so."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vdetlib_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench
bench.main()
