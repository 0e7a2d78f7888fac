"""One graphed train step (train-mode BN forward + backward + SGD) of an airplane-sized component under rocprofv3:
   cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts -o ts -- python3 tools/diag/trainstep_kernels.py"""
import sys, time, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
L, f, G, B, N = 11, 37, 128, 64, 2048
d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda().train()
opt = torch.optim.SGD(d.parameters(), lr=1e-4)
p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    z, ld = d.forward_fused(pd, gd, "inverse")
    loss = 0.5 * (ld + z * z).sum() / B
    loss.backward(); opt.step()
    return loss
for _ in range(2): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print("eager ms", (time.perf_counter() - t) / 5 * 1e3)
