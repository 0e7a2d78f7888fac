import sys, time, torch
sys.path.insert(0, ".")
from go_with_the_flows_amd import _lib
L = _lib.lib()
def t(B, G, C, f, n=200):
    pf = torch.randn(C * L.gwtf_packed_film_coupling_floats(f, G), device="cuda") * 0.05
    g = torch.randn(B, G, device="cuda")
    for _ in range(10): _lib.film_forward(g, pf, C, f, 1e-6, False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): _lib.film_forward(g, pf, C, f, 1e-6, False)
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} G={G} C={C} f={f}: {e0.elapsed_time(e1) / n * 1e3:.2f} us per call")
for cfg in [(16,128,1,64),(16,128,12,64),(32,128,12,64),(64,128,33,37),(256,128,33,37),(16,512,33,33),(16,16,1,16)]:
    t(*cfg)
