#!/usr/bin/env python3
"""Relative error of the fused SS2D core against the CPU oracle (fp32 recurrence on fp64-formed operands is the test's
business; this prints the plain oracle comparison) on a few maps.   WAVEMAMBA_HIP_LIB=<variant.so> python tools/core_accuracy.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import wave_mamba_amd as wm
from oracle import oracle
from test_gpu_parity import random_core_case

print("lib:", wm._lib.LIB_PATH)
for (B, D, H, W, N, R) in [(1, 64, 64, 64, 16, 2), (1, 64, 128, 128, 16, 2), (1, 64, 48, 40, 32, 2), (1, 64, 272, 480, 16, 2)]:
    case = random_core_case(B, D, H, W, N, R, seed=H * 100 + W)
    want = oracle.ss2d_core_raw(*case)
    got = wm.ops.ss2d_core(*[t.to("cuda:0") for t in case])
    errs = []
    for a, b in zip(got, want):
        a = a.cpu()
        errs.append((float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max())))
    print(f"B{B} D{D} {H}x{W} N{N} R{R}: " + "  ".join(f"y{i} l2 {e[0]:.2e} max {e[1]:.2e}" for i, e in enumerate(errs)))
