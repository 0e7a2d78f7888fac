"""Gradient noise at config depth: ours (HIP) and the fp32 CPU port against the fp64 CPU port, eval and train BatchNorm."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from helpers import decoder_and_state
from go_with_the_flows_amd.synth import synth_inputs
from oracle import torch_port as tp
L, f, G, B, N = 11, 37, 128, 3, 96
p, g = synth_inputs(B, N, G, 1801)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for training in (False, True):
    m, st = decoder_and_state(L, f, G, 1800)
    res = {}
    for dt in (torch.float64, torch.float32):
        tst = {k: torch.from_numpy(v).to(dt if v.dtype == np.float32 else torch.from_numpy(v).dtype).clone() for k, v in st.items()}
        for k, v in tst.items():
            if v.is_floating_point() and not k.endswith(('running_mean', 'running_var', 'eps')): v.requires_grad_(True)
        pc, gc = torch.from_numpy(p).to(dt).requires_grad_(True), torch.from_numpy(g).to(dt).requires_grad_(True)
        z, ld = tp.decoder_fused(pc, gc, tst, L, 'inverse', grad=True, training=training)
        (0.5 * (ld + z ** 2).sum() / B).backward()
        res[dt] = (pc.grad.double().numpy(), gc.grad.double().numpy(), z.detach().double().numpy())
    m = m.to('cuda:0').train(training)
    pt, gt = torch.from_numpy(p).cuda().requires_grad_(True), torch.from_numpy(g).cuda().requires_grad_(True)
    z, ld = m.forward_fused(pt, gt, 'inverse')
    (0.5 * (ld + z ** 2).sum() / B).backward()
    r64, r32 = res[torch.float64], res[torch.float32]
    print('train' if training else 'eval ', 'z: ours %.2e port32 %.2e | dp: ours %.2e port32 %.2e | dg: ours %.2e port32 %.2e  |z|max %.1f' % (
        rel(z.detach().cpu().double().numpy(), r64[2]), rel(r32[2], r64[2]), rel(pt.grad.cpu().double().numpy(), r64[0]), rel(r32[0], r64[0]),
        rel(gt.grad.cpu().double().numpy(), r64[1]), rel(r32[1], r64[1]), np.abs(r64[2]).max()))
