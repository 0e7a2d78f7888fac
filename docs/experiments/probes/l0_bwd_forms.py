import torch, time
B, C, f, G = 64, 132, 37, 128
M = C * 4 * f
g = torch.randn(B, G, device='cuda'); FS = f * G + 5 * f + f * f
arena = torch.randn(C, 2, 2, FS, device='cuda')
L0 = arena[..., :f * G].reshape(C, 2, 2, f, G)
dh = torch.randn(B, C, 2, 2, f, device='cuda')
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('fwd einsum            %.1f us' % t(lambda: torch.einsum('bg,cxhfg->bcxhf', g, L0)))
print('dL0 einsum            %.1f us' % t(lambda: torch.einsum('bcxhf,bg->cxhfg', dh, g)))
print('dL0 mm (M,B)@(B,G)    %.1f us' % t(lambda: torch.mm(dh.view(B, M).t(), g)))
print('dL0 mm  g^T form      %.1f us' % t(lambda: torch.mm(g.t(), dh.view(B, M)).t()))
print('dg einsum             %.1f us' % t(lambda: torch.einsum('bcxhf,cxhfg->bg', dh, L0)))
L0c = L0.reshape(M, G)
print('dg mm contiguous      %.1f us' % t(lambda: torch.mm(dh.view(B, M), L0c)))
print('L0 reshape copy       %.1f us' % t(lambda: L0.reshape(M, G)))
