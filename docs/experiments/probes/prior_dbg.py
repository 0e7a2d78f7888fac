import sys, numpy as np, torch
sys.path.insert(0, '.')
from go_with_the_flows_amd import prior
from go_with_the_flows_amd.synth import load_synth_
n_flows, F_, G, B = 3, 24, 16, 6
mode, training = sys.argv[1], sys.argv[2] == '1'
ref = prior.GlobalRNVPDecoder(n_flows, F_, G); load_synth_(ref, 77)
m = prior.GlobalRNVPDecoder(n_flows, F_, G); m.load_state_dict(ref.state_dict())
ref = ref.double().train(training); m = m.cuda().train(training)
rng = np.random.default_rng(5)
g = rng.standard_normal((B, G)).astype(np.float32)
n2 = 2 * n_flows
gd = torch.from_numpy(g).cuda().requires_grad_(True)
gs, mus, lvs = m(gd, mode=mode)
gt = torch.from_numpy(g).double().requires_grad_(True)
rgs, rmus, rlvs = ref(gt, mode=mode)
print('fwd err', float((torch.stack(gs).cpu().double() - torch.stack(rgs)).abs().max()), 'max |g|', float(torch.stack(rgs).abs().max()))
for which in ('gs_last', 'gs_first', 'lv_all'):
    for t in (gd, gt): t.grad = None
    m.zero_grad(); ref.zero_grad()
    if which == 'gs_last': l, rl = gs[-1].sum(), rgs[-1].sum()
    elif which == 'gs_first': l, rl = gs[0].sum(), rgs[0].sum()
    else: l, rl = sum(lvs).sum(), sum(rlvs).sum()
    l.backward(retain_graph=True); rl.backward(retain_graph=True)
    named = dict(ref.named_parameters())
    z = lambda k: named[k].grad if named[k].grad is not None else torch.zeros_like(named[k])
    errs = sorted(((float((v.grad.cpu().double() - z(k)).norm()), float(z(k).norm()), k) for k, v in m.named_parameters()), reverse=True)
    print(which, 'dg rel', float((gd.grad.cpu().double() - gt.grad).norm() / gt.grad.norm()), 'worst abs err (err, |ref|, name)', errs[:3])
