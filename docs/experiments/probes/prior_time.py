import sys, time, torch
sys.path.insert(0, '.')
from go_with_the_flows_amd import prior
from go_with_the_flows_amd.synth import load_synth_
n_flows, F_, G, B = 7, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = prior.GlobalRNVPDecoder(n_flows, F_, G); load_synth_(m, 9)
with torch.no_grad():
    for k, v in m.named_parameters():
        if 'mlp1' in k: v.mul_(0.15)
m = m.cuda().train()
g = torch.randn(B, G, device='cuda', requires_grad=True)
def step(fused):
    m._fused_ok = (lambda _g, _rows=None: True) if fused else (lambda _g, _rows=None: False)
    gs, mus, lvs = m(g, mode='inverse')
    (gs[0].square().sum() + sum(lvs).sum()).backward()
for fused in (True, False):
    for _ in range(3): step(fused)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): step(fused)
    torch.cuda.synchronize(); print('fused' if fused else 'torch', (time.perf_counter() - t) / 10 * 1e3, 'ms per fwd+bwd (eager, host-inclusive)')
