"""abs-epilogue variant of the forward kernel (f = 33..40) against the plain variant on the same FiLM records, and timing."""
import os, sys, torch
sys.path.insert(0, '.')
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
if os.environ.get("LIBV"): _lib.LIB_PATH = os.environ["LIBV"]
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
for f, L in ((37, 11), (33, 11), (40, 2), (36, 2)):
    for mode in ('inverse', 'direct'):
        d = gw.LocalCondRNVPDecoder(L, f, 128); load_synth_(d, 2); d = d.cuda().eval()
        p, g = synth_inputs(64, 2048, 128, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
        with torch.no_grad():
            z1, l1 = d.forward_fused(pd, gd, mode)
            with _lib.tuning(abs_epilogue=False):
                z0, l0 = d.forward_fused(pd, gd, mode)
        sc = z0.abs().amax(1, keepdim=True).clamp_min(1.0)
        print(f'f={f} L={L} {mode}: max |abse - plain| coords (per-point scaled) {float(((z1 - z0).abs() / sc).max()):.2e}  logdet {float((l1 - l0).abs().max()):.2e}  finite {bool(torch.isfinite(z1).all())}')
d = gw.LocalCondRNVPDecoder(11, 37, 128); load_synth_(d, 2); d = d.cuda().eval()
p, g = synth_inputs(64, 2048, 128, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
for name, ctx in (('abs-epilogue', _lib.tuning()), ('plain', _lib.tuning(abs_epilogue=False))):
    with ctx, torch.no_grad():
        pw, pf = d.engine().packed(False)
        film = _lib.film_forward(gd, pf, 33, 37, 1e-6, False)
        for _ in range(10): _lib.stack_forward(pd, pw, film, 33, 37, 0, 1e-6, 'inverse', False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): _lib.stack_forward(pd, pw, film, 33, 37, 0, 1e-6, 'inverse', False)
        e1.record(); torch.cuda.synchronize()
        print(f'{name}: stack kernel {e0.elapsed_time(e1) / 50 * 1e3:.1f} us (one component, 64 x 2048)')
