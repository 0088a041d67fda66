#!/usr/bin/env python3
"""development: run bench.py against another build of libhaslr_hip.so: tools/dev_bench_lib.py LIBDIR|- [bench.py arguments]"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import haslr_amd.hip as h
if sys.argv[1] != "-":
    h._LIBDIR = os.path.join(root, sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
