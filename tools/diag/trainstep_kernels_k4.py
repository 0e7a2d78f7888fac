"""One eager train step (train-mode BN forward + backward + SGD) of the airplane config's FOUR decoders through the K-batched pipeline,
for rocprofv3 (tools/pmc_any.sh): the K = 4 launches the whole-model step runs, without the model around them."""
import sys, time, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
K, L, f, G, B, N = 4, 11, 37, 128, 64, 2048
decs = []
for k in range(K):
    d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2 + k); decs.append(d.cuda().train())
stack = gw.MixtureStack(decs)
opt = torch.optim.SGD([p for d in decs for p in d.parameters()], lr=1e-4)
p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    z, ld = stack.forward_all(pd, gd, "inverse")
    loss = 0.5 * (ld + z * z).sum() / B
    loss.backward(); opt.step()
    return loss
for _ in range(2): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); print("eager ms", (time.perf_counter() - t) / 3 * 1e3)
