import sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import decoder_and_state
from go_with_the_flows_amd.synth import synth_inputs
L, f, G = 11, 37, 128
B, N = int(sys.argv[1]), int(sys.argv[2])
m, _ = decoder_and_state(L, f, G, 31); m = m.cuda().train()
if len(sys.argv) > 3: m.engine().force_autograd_chain = True
p, g = synth_inputs(B, N, G, 32); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
targets = [m.flows[0].nvp2.T_mu_0[3].weight, m.flows[5].nvp1.T_logvar_0[3].weight, m.flows[10].nvp3.T_mu_0[3].weight]
gen = torch.Generator().manual_seed(5)
dirs = [torch.randn(t.shape, generator=gen).cuda() for t in targets]
def loss_fn():
    state = {k: v.clone() for k, v in m.state_dict().items() if 'running' in k or 'num_batches' in k}
    z, ld = m.forward_fused(pd, gd, 'inverse')
    m.load_state_dict(state, strict=False)
    return (0.5 * (ld + z * z).sum(dim=(1, 2)) / N).mean()
loss = loss_fn(); loss.backward()
print("analytic per target", [float((t.grad * d).sum()) for t, d in zip(targets, dirs)])
base = [t.detach().clone() for t in targets]
for h in (2e-3, 5e-4, 1e-4):
    out = []
    for k in range(3):
        vals = []
        for sgn in (+1, -1):
            with torch.no_grad():
                for t, b0 in zip(targets, base): t.copy_(b0)
                targets[k].add_(dirs[k], alpha=sgn * h)
            vals.append(float(loss_fn()))
        out.append((vals[0] - vals[1]) / (2 * h))
    print("h", h, "fd per target", out)
