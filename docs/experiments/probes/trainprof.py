import sys, time, torch, cProfile, pstats
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
L, f, G, B, N = 4, 64, 128, 32, 2048
d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda().train()
p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
with torch.no_grad():
    for _ in range(3): d.forward_fused(pd, gd, "inverse")
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): d.forward_fused(pd, gd, "inverse")
    torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
