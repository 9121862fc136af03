#!/usr/bin/env python3
"""`python3 global_optimization_hip.py <body_path> <fit_path> <mode>` -- the reference's command
line (global_optimization.py:658-660) on the MI355X path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fdcap_amd  # noqa: E402,F401
from fdcap_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
