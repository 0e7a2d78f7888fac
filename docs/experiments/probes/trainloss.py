import sys, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
L, f, G, B, N = 11, 37, 128, 64, 2048
d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda().train()
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
opt = torch.optim.SGD(d.parameters(), lr=lr)
p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
for i in range(12):
    opt.zero_grad(set_to_none=True)
    z, ld = d.forward_fused(pd, gd, "inverse")
    loss = 0.5 * (ld + z * z).sum() / (B * N)
    loss.backward()
    gn = torch.sqrt(sum((q.grad ** 2).sum() for q in d.parameters()))
    opt.step()
    print(i, float(loss), float(gn))
