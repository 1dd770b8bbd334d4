#!/usr/bin/env python3
"""Builds tools/libfqsx_timing.so: the product sources with -DFQSX_TIMING (in-kernel section timers and the per-launch
role time stamps read by tools/gpu_timing.py / tools/gpu_roles.py).  Diagnostic only; never loaded by the package."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
print(g.build_hip(True, extra_flags=["-DFQSX_TIMING"] + (["-DFQSX_TIMING_MODELS"] if "models" in sys.argv[1:] else []) + (["-DFQSX_TIMING_PE"] if "pe" in sys.argv[1:] else []), out=os.path.join(ROOT, "tools", "libfqsx_timing.so")))
