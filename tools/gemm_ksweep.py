"""K-sweep of the tile-GEMM core (KC x KC operands, like the Cholesky trailing update) on the GPU box."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bobe_amd import _lib  # noqa: E402

lib = _lib.load()
M = N = 4096
for K in (128, 256, 384, 512, 1024, 4096):
    A = torch.randn((M, K), dtype=torch.float64, device="cuda")
    B = torch.randn((N, K), dtype=torch.float64, device="cuda")
    Cm = torch.empty((M, N), dtype=torch.float64, device="cuda")
    args = (0, 0, 0, M, N, K, _lib.ptr(A), K, _lib.ptr(B), K, _lib.ptr(Cm), N)
    lib.bobe_debug_gemm(*args)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.bobe_debug_gemm(*args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"gemm {M}x{N}x{K}: {dt * 1e6:.1f} us  {2.0 * M * N * K / dt / 1e12:.1f} TFLOP/s", flush=True)
