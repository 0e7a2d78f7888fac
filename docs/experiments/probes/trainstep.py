import sys, time, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
for (L, f, G, B, N) in [(4, 64, 128, 32, 2048), (11, 37, 128, 64, 2048)]:
    d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda().train()
    opt = torch.optim.SGD(d.parameters(), lr=1e-4)
    p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    def step():
        opt.zero_grad(set_to_none=True)
        z, ld = d.forward_fused(pd, gd, "inverse")
        loss = 0.5 * (ld + z * z).sum() / B
        loss.backward(); opt.step()
    for _ in range(2): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"train step (fwd train-BN + bwd + SGD) L={L} f={f} B={B} N={N}: {dt*1e3:.1f} ms  ({B*N/dt/1e6:.2f} Mpts/s)")
    d.eval()
    def estep():
        opt.zero_grad(set_to_none=True)
        z, ld = d.forward_fused(pd, gd, "inverse")
        (0.5 * (ld + z * z).sum() / B).backward()
    for _ in range(2): estep()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): estep()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"   eval-BN fwd+bwd: {dt*1e3:.1f} ms")
