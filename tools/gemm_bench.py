"""Micro-benchmarks on the GPU box: fp64 MFMA issue peak and the tile-GEMM core's steady state."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib  # noqa: E402

lib = _lib.load()
for w in (1, 2):
    t = C.c_double()
    lib.bobe_debug_mfma_peak(0, w, C.byref(t))
    print(f"mfma f64 16x16x4 issue peak, {w} wave(s)/SIMD: {t.value:.1f} TFLOP/s", flush=True)

for (M, N, K) in [(4096, 4096, 4096), (2048, 2048, 8192), (8192, 8192, 512), (4096, 4096, 128)]:
    for la in (0, 1):
        for lb in (0, 1):
            A = torch.randn((M, K) if la == 0 else (K, M), dtype=torch.float64, device="cuda")
            B = torch.randn((N, K) if lb == 0 else (K, N), dtype=torch.float64, device="cuda")
            Cm = torch.empty((M, N), dtype=torch.float64, device="cuda")
            args = (0, la, lb, M, N, K, _lib.ptr(A), A.shape[1], _lib.ptr(B), B.shape[1], _lib.ptr(Cm), N)
            lib.bobe_debug_gemm(*args)
            torch.cuda.synchronize()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.bobe_debug_gemm(*args)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"gemm {M}x{N}x{K} la={la} lb={lb}: {dt * 1e3:.3f} ms  {2.0 * M * N * K / dt / 1e12:.1f} TFLOP/s", flush=True)
