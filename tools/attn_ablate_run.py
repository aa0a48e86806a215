#!/usr/bin/env python
"""Runs tools/attn_bench.py once per tool-only attention build (tools/attn_ablate.sh), one process each."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sorted(glob.glob(os.path.join(ROOT, "edgestyle_amd", "lib", "ablate", "libes_attn_*.so"))):
    print("==", os.path.basename(lib), flush=True)
    env = dict(os.environ, ES_HIP_LIB=lib, ATTN_BENCH_SHAPES=os.environ.get("ATTN_BENCH_SHAPES", "0,2,4"))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_bench.py")], env=env, check=False)
