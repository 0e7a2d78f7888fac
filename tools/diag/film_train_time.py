"""FiLM heads of the train step (autograd._film_train): forward + backward of all K x C couplings' heads on strided views of one
arena, graph-replayed."""
import sys, torch
sys.path.insert(0, '.')
from go_with_the_flows_amd import autograd as ag
C, f, G, B = 132, 37, 128, 64
dev = 'cuda'
torch.manual_seed(0)
FS = f * G + 5 * f + f * f
raw = (torch.randn(C, 2, 2 * FS + 12, device=dev) * 0.1).requires_grad_(True)
o_hbn, o_l1, o_b1 = f * G, f * G + 4 * f, f * G + 4 * f + f * f
zero = torch.zeros(C, 2, 8, device=dev)


def views():       # per step, as the engine does (autograd nodes of a previous step must not be reused inside a capture)
    films = raw[:, :, 7:7 + 2 * FS].reshape(C, 2, 2, FS)
    return {'raw': zero, 'L0': films[..., :o_hbn].reshape(C, 2, 2, f, G),
            'hbn': list(films[..., o_hbn:o_l1].reshape(C, 2, 2, 4, f).unbind(3)), 'L1': films[..., o_l1:o_b1].reshape(C, 2, 2, f, f),
            'b1': films[..., o_b1:]}


g = torch.randn(B, G, device=dev, requires_grad=True)
wa, wb = torch.randn(B, C, 2, f, device=dev), torch.randn(B, C, 2, f, device=dev)
for name, fn in (('current', ag._film_train),):
    def step():
        raw.grad = None; g.grad = None
        a, bsh, mean, var = fn(views(), g, 1e-6)
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
    print(f'FiLM heads train fwd+bwd ({name}), C={C} f={f} G={G} B={B}: {e0.elapsed_time(e1) * 10:.1f} us per replay')
