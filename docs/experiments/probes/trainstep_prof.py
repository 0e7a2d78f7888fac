import sys, time, torch, cProfile, pstats
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
L, f, G, B, N = 4, 64, 128, 32, 2048
d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda().train()
opt = torch.optim.SGD(d.parameters(), lr=1e-4)
p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    z, ld = d.forward_fused(pd, gd, "inverse")
    loss = 0.5 * (ld + z * z).sum() / B
    loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
