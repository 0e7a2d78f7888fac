import sys, time, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
for (L, f, G, B, N) in [(4, 64, 128, 32, 2048), (11, 37, 128, 64, 2048)]:
    d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda()
    p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    for mode_name, m in (("eval", d.eval()), ("train", d.train())):
        with torch.no_grad():
            for _ in range(3): m.forward_fused(pd, gd, "inverse")
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): m.forward_fused(pd, gd, "inverse")
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print(f"L={L} f={f} B={B} N={N} {mode_name}: {dt*1e3:.3f} ms  ({B*N/dt/1e6:.1f} Mpts/s)")
