#!/usr/bin/env python3
"""A/B of the training step for kernel variants built as separate libraries:

    TSDIFF_LIB=/path/to/variant.so python tools/ab_train.py [steps]

Prints ms/step of `bench.py --workload train` (a new batch every step) for the selected library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib  # noqa: E402

if os.environ.get("TSDIFF_LIB"):
    _lib.LIB_PATH = os.environ["TSDIFF_LIB"]
import bench  # noqa: E402

steps = sys.argv[1] if len(sys.argv) > 1 else "40"
sys.argv = ["bench.py", "--workload", "train", "--steps", steps, "--warmup", "8"]
bench.main()
