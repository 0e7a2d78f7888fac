"""FiLM heads of the train step (autograd._film_train): forward + backward of all K x C couplings' heads, graph-replayed."""
import sys, torch
sys.path.insert(0, '.')
from go_with_the_flows_amd import autograd as ag
C, f, G, B = 132, 37, 128, 64
dev = 'cuda'
torch.manual_seed(0)
raw = torch.zeros(C, 2, 8, device=dev)
L0 = (torch.randn(C, 2, 2, f, G, device=dev) * 0.1).requires_grad_(True)
hbn = [torch.rand(C, 2, 2, f, device=dev).add_(0.5).requires_grad_(True), torch.randn(C, 2, 2, f, device=dev).mul_(0.1).requires_grad_(True),
       torch.zeros(C, 2, 2, f, device=dev), torch.ones(C, 2, 2, f, device=dev)]
L1 = (torch.randn(C, 2, 2, f, f, device=dev) * 0.1).requires_grad_(True)
b1 = torch.zeros(C, 2, 2, f, device=dev, requires_grad=True)
g = torch.randn(B, G, device=dev, requires_grad=True)
P = {'raw': raw, 'L0': L0, 'hbn': hbn, 'L1': L1, 'b1': b1}
wa, wb = torch.randn(B, C, 2, f, device=dev), torch.randn(B, C, 2, f, device=dev)

def step():
    for t in (L0, hbn[0], hbn[1], L1, b1, g):
        t.grad = None
    a, bsh, mean, var = ag._film_train(P, g, 1e-6)
    ((a * wa).sum() + (bsh * wb).sum()).backward()

s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr): step()
for _ in range(10): gr.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): gr.replay()
e1.record(); torch.cuda.synchronize()
print(f'FiLM heads train fwd+bwd, C={C} f={f} G={G} B={B}: {e0.elapsed_time(e1) * 10:.1f} us per replay')
