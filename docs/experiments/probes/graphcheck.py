import sys, copy, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
L, f, G, B, N = 2, 19, 16, 4, 256
def make():
    d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); return d.cuda().train()
p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
def mk_step(d, opt):
    def step():
        opt.zero_grad(set_to_none=True)
        z, ld = d.forward_fused(pd, gd, "inverse")
        loss = 0.5 * (ld + z * z).sum() / (B * N)
        loss.backward(); opt.step()
        return loss
    return step
d1 = make(); o1 = torch.optim.SGD(d1.parameters(), lr=1e-3); s1 = mk_step(d1, o1)
eager = [s1().item() for _ in range(6)]
d2 = make(); o2 = torch.optim.SGD(d2.parameters(), lr=1e-3); s2 = mk_step(d2, o2)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    warm = [s2().item() for _ in range(3)]
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph(); o2.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    loss = s2()
vals = []
for _ in range(3):
    graph.replay(); vals.append(loss.item())
print("eager :", ["%.6f" % v for v in eager])
print("graph :", ["%.6f" % v for v in warm + vals])
