"""Relative cost of one resident round of workgroups at 16 / 32 / 64 points per wave (calibrates the tile chooser)."""
import sys, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
for (L, f) in [(11, 37), (4, 64), (6, 19)]:
    d = gw.LocalCondRNVPDecoder(L, f, 128); load_synth_(d, 2); d = d.cuda().eval()
    p, g = synth_inputs(64, 2048, 128, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    res = {}
    for ppw in (16, 32, 64):
        _lib.set_tuning(ppw)
        with torch.no_grad():
            for _ in range(5): d.forward_fused(pd, gd, "inverse")
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): d.forward_fused(pd, gd, "inverse")
            e1.record(); torch.cuda.synchronize()
        res[ppw] = e0.elapsed_time(e1) / 50
    _lib.set_tuning(0)
    t64 = res[64]
    print(f'L={L} f={f}: ms per call ppw16 {res[16]:.4f} (4 rounds) ppw32 {res[32]:.4f} (2 rounds) ppw64 {res[64]:.4f} (1 round)'
          f'  ->  w(1) = {res[16]/4/t64:.2f}  w(2) = {res[32]/2/t64:.2f}')
