"""Run-to-run deviation of the K-batched train gradients at a size where the light pass takes its 256-point tile, with either
tile (argument `small`: the per-call tuning word GWTF_TUNE_SMALL_LIGHT_TILE forces 128 points): are the two tiles equally (ir)reproducible?  (ReLU kinks, docs/LOG.md 4.11.)"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import go_with_the_flows_amd as gw
from test_gpu_parity import decoder_and_state, synth_inputs, dev, DEV
L, f, G, B, N = 1, 8, 16, 64, 2048
decs = [decoder_and_state(L, f, G, 640 + k)[0].to(DEV).train() for k in range(2)]
p, g = synth_inputs(B, N, G, 641)
gen = torch.Generator().manual_seed(642)
wz, wl = torch.randn(2, B, 3, N, generator=gen).to(DEV), torch.randn(2, B, 3, N, generator=gen).to(DEV)
signed = 'signed' in sys.argv
if 'small' in sys.argv:
    from go_with_the_flows_amd import _lib
    _lib.set_tuning(_lib.TUNE_SMALL_LIGHT_TILE)
if not signed:
    wz, wl = wz.abs(), wl.abs()
state = [{k: v.clone() for k, v in d.state_dict().items()} for d in decs]
runs = []
for rep in range(24):
    for d, s in zip(decs, state):
        d.load_state_dict(s); d.zero_grad(set_to_none=True)
    pd, gd = dev(p).requires_grad_(True), dev(g).requires_grad_(True)
    z, ld = gw.MixtureStack(decs).forward_all(pd, gd, 'inverse')
    ((z * wz).sum() + (ld * wl).sum()).backward()
    runs.append([torch.cat([q.grad.reshape(-1) for q in d.parameters()]) for d in decs])
for k in range(2):
    dv = sorted(float((r[k] - runs[0][k]).norm() / runs[0][k].norm()) for r in runs[1:])
    print('decoder', k, 'relative deviation from run 0: median %.1e  max %.1e  runs above 1e-5: %d of %d' % (dv[len(dv) // 2], dv[-1], sum(x > 1e-5 for x in dv), len(dv)))
