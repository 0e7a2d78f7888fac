import sys, time, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
for (L, f, G, B, N) in [(4, 64, 128, 32, 2048), (11, 37, 128, 64, 2048)]:
    d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda().train()
    opt = torch.optim.SGD(d.parameters(), lr=1e-9)
    p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    def step():
        opt.zero_grad(set_to_none=True)
        z, ld = d.forward_fused(pd, gd, "inverse")
        loss = 0.5 * (ld + z * z).sum() / B
        loss.backward(); opt.step()
        return loss
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = step()
    l0 = None
    for _ in range(3): graph.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): graph.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"graphed train step L={L} f={f} B={B} N={N}: {dt*1e3:.2f} ms ({B*N/dt/1e6:.1f} Mpts/s) loss {loss.item():.4f}")
