"""Intermittent two-valued gradients?  K decoders through the K-batched train pipeline, the same step many times."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import go_with_the_flows_amd as gw
from helpers import decoder_and_state
from go_with_the_flows_amd.synth import synth_inputs
K, L, f, G, B, N = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (3, 2, 8, 16, 4, 48))]
decs = [decoder_and_state(L, f, G, 20 + k)[0].cuda().train() for k in range(K)]
ms = gw.MixtureStack(decs)
p, g = synth_inputs(B, N, G, 11)
rng = np.random.default_rng(12)
wz, wl = (torch.from_numpy(rng.normal(size=(K, B, 3, N)).astype(np.float32)).cuda() for _ in range(2))
res = []
for rep in range(60):
    for d in decs:
        d.zero_grad(set_to_none=True)
    pt, gt = torch.from_numpy(p).cuda().requires_grad_(True), torch.from_numpy(g).cuda().requires_grad_(True)
    z, ld = ms.forward_all(pt, gt, 'inverse')
    ((z * wz).sum() + (ld * wl).sum()).backward()
    res.append((z.detach().clone(), pt.grad.clone(), gt.grad.clone(), [torch.cat([q.grad.reshape(-1) for q in d.parameters()]) for d in decs]))
ref = res[0]
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-20))
dev = []
for r in res[1:]:
    dev.append((rel(r[0], ref[0]), rel(r[1], ref[1]), rel(r[2], ref[2]), [rel(a, b) for a, b in zip(r[3], ref[3])]))
big = [i for i, d in enumerate(dev) if max(d[1], d[2], *d[3]) > 1e-5]
print('config', (K, L, f, G, B, N), 'deviating runs', len(big), 'of', len(dev))
for i in big[:3]:
    print('  run', i + 1, 'z %.1e dp %.1e dg %.1e per-decoder' % dev[i][:3], ['%.1e' % v for v in dev[i][3]])
