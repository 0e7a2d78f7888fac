"""Does a replayed hipGraph run independent branches (captured on forked streams) concurrently?  Chains of tiny dependent
kernels (latency-bound, one workgroup each) on 1 / 2 / 4 forked streams inside one graph."""
import time, torch
dev = torch.device('cuda')
def build(nbranch, nk=300):
    xs = [torch.randn(256, device=dev) for _ in range(nbranch)]
    side = [torch.cuda.Stream() for _ in range(nbranch)]
    def work():
        cur = torch.cuda.current_stream()
        for s, x in zip(side, xs):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(nk):
                    x.mul_(1.0001)
        for s in side:
            cur.wait_stream(s)
    warm = torch.cuda.Stream(); warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        work()
    torch.cuda.current_stream().wait_stream(warm); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work()
    return g
for nb in (1, 2, 4):
    g = build(nb)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f'{nb} branch(es) x 300 dependent tiny kernels: {dt*1e3:.3f} ms per replay ({dt/300*1e6:.2f} us per kernel slot)')
