import os, sys, torch
sys.path.insert(0, '.')
from go_with_the_flows_amd import _lib
if os.environ.get('LIBV'):
    _lib.LIB_PATH = os.environ['LIBV']
from go_with_the_flows_amd import encoders
m = encoders.FeatureEncoder(1, 512, 128).cuda().train()
m.hip_max_width = 4096
x = torch.randn(64, 512, device='cuda', requires_grad=True)
for _ in range(5):
    m.zero_grad(set_to_none=True); x.grad = None
    mu, lv = m(x); ((mu * mu).sum() + (lv * lv).sum()).backward()
torch.cuda.synchronize()
