#!/usr/bin/env python3
"""Dev: cProfile of the host side of the C2 training step of tools/bench_full_step_c2.py (a) (sparse_grad: fused).  Prints the functions by cumulative time."""
import cProfile, os, pstats, runpy, sys
sys.argv = [sys.argv[0]]
os.environ["ONLY_A"] = "1"
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_full_step_c2.py"), run_name="prof")
