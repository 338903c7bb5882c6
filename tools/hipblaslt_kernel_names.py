#!/usr/bin/env python
"""Measurement tool (tools/ only): launches torch.nn.functional.linear on the LLaMA GEMM shapes so that a rocprofv3 --kernel-trace of
this script shows WHICH hipBLASLt kernels (macro tile, MFMA shape, wave tiling, LDS buffering are spelled out in their names)
the yardstick of tools/gemm_yardstick.py ran.  Nothing in the product links hipBLASLt."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
for M, N, K in [(20480, 12288, 4096), (20480, 4096, 4096), (20480, 22016, 4096), (20480, 4096, 11008), (18464, 4096, 1024), (73760, 3072, 768)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    for _ in range(5):
        F.linear(a, w)
    torch.cuda.synchronize()
