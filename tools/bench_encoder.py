#!/usr/bin/env python3
"""Time the fused PointNet encoder against the library-GEMM path at the airplane batch (64 x 2048).  GPU box only."""
import sys
import time
import torch
sys.path.insert(0, '.')
from go_with_the_flows_amd import encoders
from go_with_the_flows_amd.synth import load_synth_, synth_inputs

B, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 2048
m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
load_synth_(m, 1)
m = m.cuda().eval()
x = torch.from_numpy(synth_inputs(B, N, 4, 2)[0]).cuda()


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


flops = 2 * (3 * 64 + 64 * 128 + 128 * 256 + 256 * 512) * B * N
with torch.no_grad():
    a = t(lambda: m.forward_max(x))
    b = t(lambda: m(x))
    c = t(lambda: torch.max(m.features(x), dim=2)[0])
print(f'fused pooled  : {a:.3f} ms  {flops / a / 1e9:.1f} TFLOP/s (algorithmic fp32)  {B * N / a / 1e3:.1f} Mpts/s')
print(f'fused features: {b:.3f} ms')
print(f'library path  : {c:.3f} ms  (torch.matmul + batch_norm + relu + max)')
