import sys, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
from torch.profiler import profile, ProfilerActivity
L, f, G, B, N = 11, 37, 128, 8, 512
d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda().train()
opt = torch.optim.SGD(d.parameters(), lr=1e-6)
p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    z, ld = d.forward_fused(pd, gd, "inverse")
    loss = 0.5 * (ld + z * z).sum() / B
    loss.backward(); opt.step()
for _ in range(2): step()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:25]: print(f"{e.count:6d}  {e.key}")
