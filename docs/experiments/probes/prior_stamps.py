"""Phase timeline of ONE flow step of the prior flow's backward kernel (debug build with -DGWTF_DBG_PRIOR_STAMPS: thread 0 stamps
wall_clock64 at the phase boundaries of the second step into an unused BatchNorm-buffer slot of the gradient arena).
Build the debug library first (build container):
  cd go_with_the_flows_amd/csrc && hipcc <CXXFLAGS of the Makefile> -DGWTF_DBG_PRIOR_STAMPS -c gwtf_prior.hip -o /tmp/prior_dbg.o && \
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/diag/_dbg/lib_prior.so $(ls *.o | grep -v gwtf_prior.o) /tmp/prior_dbg.o
Round-3 reading (64 rows, G = F = 128, per flow step): affine 8 us, hidden recompute 83, dH / dW1 54, Swish + BatchNorm 20, dkept / dW0 46."""
import os, sys, torch
sys.path.insert(0, '.')
from go_with_the_flows_amd import _lib
_lib.LIB_PATH = os.environ.get('LIBV', 'tools/diag/_dbg/lib_prior.so')
from go_with_the_flows_amd import prior
from go_with_the_flows_amd.synth import load_synth_
n_flows, F_, G, B = 7, 128, 128, 64
m = prior.GlobalRNVPDecoder(n_flows, F_, G); load_synth_(m, 9); m = m.cuda().train()
g = torch.randn(B, G, device='cuda')
L = _lib.lib()
raw = m._raw_arena().detach().contiguous()
st_ = torch.cuda.current_stream().cuda_stream
lists = torch.empty(3, 14, B, G, device='cuda'); ws = torch.empty(L.gwtf_prior_workspace_floats(B, G, F_), device='cuda'); st = torch.zeros(14, 2, 2, F_, device='cuda')
_lib.check(L.gwtf_prior_forward(g.data_ptr(), raw.data_ptr(), lists[0].data_ptr(), lists[1].data_ptr(), lists[2].data_ptr(), ws.data_ptr(), st.data_ptr(), 7, B, G, F_, 1e-6, 1, 1, st_))
g_gs = torch.randn(14, B, G, device='cuda'); g_lv = torch.randn(14, B, G, device='cuda'); g_raw = torch.zeros_like(raw); g_g = torch.empty(B, G, device='cuda')
for _ in range(3):
    g_raw.zero_()
    _lib.check(L.gwtf_prior_backward(g.data_ptr(), raw.data_ptr(), lists[0].data_ptr(), lists[1].data_ptr(), lists[2].data_ptr(), g_gs.data_ptr(), g_lv.data_ptr(), ws.data_ptr(), g_raw.data_ptr(), g_g.data_ptr(), 7, B, G, F_, 1e-6, 1, 1, st_))
torch.cuda.synchronize()
off = L.gwtf_prior_raw_offset(7, G, F_, 13) + F_ * 64 + 2 * F_
stamps = g_raw[off:off + 16].view(torch.int64).cpu().tolist()[:7]
names = ['B1 affine', 'B2 hidden recompute', 'B3 dH/dW1/db1', 'B4 swish+BN bwd', 'B5 dkept/dW0', 'B6 combine']
for n, a, b in zip(names, stamps, stamps[1:]):
    print(f'{n:22s} {(b - a) / 100.0:8.1f} us')
print('step total', (stamps[6] - stamps[0]) / 100.0, 'us')
