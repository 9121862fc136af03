#!/usr/bin/env python3
"""`python3 optimization_hip.py <gen_path> <fit_path>` -- the reference's per-frame smoother command line
(optimization.py:297-349) on the MI355X path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fdcap_amd  # noqa: E402,F401
from fdcap_amd.smoother import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
