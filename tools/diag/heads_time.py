"""Per-shape heads: HIP layer kernels (csrc/gwtf_heads.hip) against the library modules, forward + backward, graph-replayed."""
import sys, torch
sys.path.insert(0, '.')
from go_with_the_flows_amd import encoders
for name, cls, ctor, kw in (('g_posterior', encoders.FeatureEncoder, (1, 512, 128), dict(deterministic=False)),
                            ('p_prior', encoders.FeatureEncoder, (1, 128, 3), dict(deterministic=False)),
                            ('p_prior512', encoders.FeatureEncoder, (1, 512, 3), dict(deterministic=False)),
                            ('weights512', encoders.WeightsEncoder, (3, 512, 4), dict(deterministic=True)),
                            ('weights', encoders.WeightsEncoder, (3, 128, 4), dict(deterministic=True))):
    for B in (8, 64):
        res = {}
        for hip in (True, False):
            torch.manual_seed(0)
            m = cls(*ctor, **kw).cuda().train()
            m.hip_max_width = 4096
            if not hip:
                m._hip_layers = lambda x: None
                m._head = lambda seq, h, act=0: (torch.nn.functional.log_softmax(seq(h), dim=1) if act == 2 else seq(h))
            x = torch.randn(B, ctor[1], device='cuda', requires_grad=True)

            def step():
                m.zero_grad(set_to_none=True)
                x.grad = None
                y = m(x)
                ys = y if isinstance(y, tuple) else (y,)
                sum((o * o).sum() for o in ys).backward()
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3): step()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): step()
            for _ in range(10): g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): g.replay()
            e1.record(); torch.cuda.synchronize()
            res[hip] = e0.elapsed_time(e1) * 10
        print(f'{name:12s} B={B:3d}  fwd+bwd per replay: HIP {res[True]:7.1f} us   library {res[False]:7.1f} us')
