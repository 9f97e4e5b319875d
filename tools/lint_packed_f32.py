#!/usr/bin/env python3
"""The ISA lint lives in the package (wave_mamba_amd/_lint_packed_f32.py: the build needs it wherever the package is
deployed); this is its command-line entry and the name older scripts import.

usage: python tools/lint_packed_f32.py [path/to/lib.so]     exit status 1 when an offending instruction exists"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from wave_mamba_amd._lint_packed_f32 import *          # noqa: F401,F403
from wave_mamba_amd._lint_packed_f32 import main, LLVM, disassemble, offending      # noqa: F401

if __name__ == "__main__":
    sys.exit(main())
