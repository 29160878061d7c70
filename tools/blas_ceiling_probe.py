#!/usr/bin/env python3
"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) sustains on this box for the shapes of the per-pair layers,
as an independent measure of the f16 / bf16 matrix-core rate the part delivers under its power limit.  Measuring stick
only: nothing in the product calls a BLAS library.

    python tools/blas_ceiling_probe.py"""
import time

import torch


def bench(m, n, k, dtype, iters=20, relu_sparse=False):
    a = torch.randn(m, k, device='cuda', dtype=torch.float32)
    if relu_sparse:
        a = torch.relu(a)                      # half of the entries exactly zero, like the layer's real operand
    a = a.to(dtype)
    b = (torch.randn(k, n, device='cuda', dtype=torch.float32) * 0.03).to(dtype)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * m * n * k / (ms * 1e-3) / 1e12


def main():
    print(torch.__version__, torch.cuda.get_device_name(0))
    # per-pair layer, its half, the layer-0 product (16000 rx preambles x 10240 samples), the regressor, two library-friendly squares
    for (m, n, k) in ((512000, 1024, 1024), (262144, 1024, 1024), (16000, 1024, 10240), (512000, 256, 1024), (8192, 8192, 8192), (16384, 16384, 4096)):
        for dtype in (torch.float16, torch.bfloat16):
            for sparse in (False, True):
                ms, tf = bench(m, n, k, dtype, relu_sparse=sparse)
                print('M=%7d N=%5d K=%5d %-8s %-12s %8.3f ms  %7.1f TFLOP/s  (%.2f of 2500)'
                      % (m, n, k, str(dtype).split('.')[-1], 'relu operand' if sparse else 'dense', ms, tf, tf / 2500), flush=True)
    # long run: the sustained rate after the clocks have settled
    m, n, k = 512000, 1024, 1024
    a = torch.relu(torch.randn(m, k, device='cuda')).half()
    b = (torch.randn(k, n, device='cuda') * 0.03).half()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 0
    while time.perf_counter() - t0 < 5.0:
        for _ in range(50):
            c = a @ b
        torch.cuda.synchronize()
        it += 50
    dt = time.perf_counter() - t0
    print('5 s of back-to-back f16 GEMMs (M=512000, relu operand): %.1f TFLOP/s' % (2.0 * m * n * k * it / dt / 1e12))


if __name__ == '__main__':
    main()
