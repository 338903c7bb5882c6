#!/usr/bin/env python
"""Is the bf16 GEMM plateau a POWER plateau?  (round 5; measurement tool, tools/ only)

Runs one GEMM shape back to back for a few seconds per arm and samples the board's power and shader clock with rocm-smi in a side
thread.  Arms: {gemm256 through the C-ABI, hipBLASLt through torch} x {N(0,1) operands, zero operands}.  If the kernel is bound by
the power cap, the random-operand arms sit at the cap with a REDUCED shader clock and the zero-operand arms run at a higher clock and
proportionally more TFLOP/s with the same instruction stream.
usage: python tools/power_probe.py > profiles/r05_power_probe.txt"""
import ctypes
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
M, N, K = 20480, 12288, 4096            # the LLaMA qkv projection of a 32-crop batch


def smi_sample():
    """(power W, sclk MHz) from one rocm-smi call; None where the field is missing."""
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
        pw = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[\d.]+$", str(v))), None)
        ck = next((v for k, v in card.items() if "sclk" in k.lower()), None)
        mhz = float(re.search(r"(\d+)\s*Mhz", str(ck), re.I).group(1)) if ck and re.search(r"(\d+)\s*Mhz", str(ck), re.I) else None
        return pw, mhz
    except Exception:
        return None, None


def arm(name, fn, seconds=4.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi_sample())
            time.sleep(0.2)
    th = threading.Thread(target=sampler)
    th.start()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / n
    pw = [s[0] for s in samples[2:] if s[0] is not None]
    ck = [s[1] for s in samples[2:] if s[1] is not None]
    print(f"{name:34s} {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.0f} TFLOP/s   power {sum(pw) / max(len(pw), 1):6.0f} W (max {max(pw or [0]):.0f})   "
          f"sclk {sum(ck) / max(len(ck), 1):5.0f} MHz (min {min(ck or [0]):.0f}, max {max(ck or [0]):.0f})   [{len(pw)} samples]", flush=True)


print(f"# {torch.cuda.get_device_name(0)}; shape M {M} N {N} K {K} bf16; idle sample: power/sclk = {smi_sample()}")
for data in ("random", "zeros"):
    if data == "random":
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    else:
        a = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def eng():
        assert lib.vstar_op_gemm(None, P(a), K, P(w), None, None, N, P(c), N, 0, M, N, K, 0 | 0x100 | _lib.EPI_TILE256) == 0

    def eng4():      # round 6: the 4-wave / AGPR kernel (gemm4w.hip)
        assert lib.vstar_op_gemm(None, P(a), K, P(w), None, None, N, P(c), N, 0, M, N, K, 0 | 0x100 | _lib.EPI_TILE4W) == 0

    tag = os.environ.get("POWER_PROBE_TAG", "")
    arm(f"gemm256{tag:<10s}{data}", eng)
    arm(f"gemm4w {tag:<10s}{data}", eng4)
    if not tag:
        arm(f"hipBLASLt {data}", lambda: F.linear(a, w))
