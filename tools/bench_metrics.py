#!/usr/bin/env python3
"""Time the structural-loss kernels at evaluate_ae sizes (b clouds of 2048 points).  GPU box only."""
import sys
import time
import torch
sys.path.insert(0, '.')
from go_with_the_flows_amd import metrics

b, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 2048
x = torch.randn(b, n, 3, device='cuda') * 0.3
y = torch.randn(b, n, 3, device='cuda') * 0.3


def t(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ms = t(lambda: metrics.nn_distance_raw(x, y), 20)
print(f'nn_distance  b={b} n={n}: {ms:.3f} ms  ({2 * b * n * n / ms / 1e6:.1f} Gpair/s)')
ms = t(lambda: metrics.approx_match(x, y), 3)
print(f'approx_match b={b} n={n}: {ms:.3f} ms  ({27 * b * n * n / ms / 1e6:.1f} Gpair-sweeps/s)')
ms = t(lambda: metrics.match_cost(x, y), 3)
print(f'match_cost fused (emd_approx path) b={b}: {ms:.3f} ms  ({27 * b * n * n / ms / 1e6:.1f} Gpair-sweeps/s)')
