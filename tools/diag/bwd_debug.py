import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import coupling_and_state, decoder_and_state
from go_with_the_flows_amd.synth import synth_inputs
from oracle import torch_port as tp
import go_with_the_flows_amd as gw
def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
f, G, B, N = 8, 8, 2, 20
for pi, warp in enumerate(gw.WARP_PATTERNS):
    m, st = coupling_and_state(f, G, warp, 10 + pi)
    m = m.cuda().eval()
    p, g = synth_inputs(B, N, G, 3)
    rng = np.random.default_rng(1)
    wz, wl = rng.normal(size=(B, 3, N)).astype(np.float32), rng.normal(size=(B, 3, N)).astype(np.float32)
    tst = {k: torch.from_numpy(v).clone().requires_grad_(v.dtype == np.float32 and not k.endswith(("running_mean", "running_var", "eps"))) for k, v in st.items()}
    pc, gc = torch.from_numpy(p).requires_grad_(True), torch.from_numpy(g).requires_grad_(True)
    with torch.enable_grad():
        zc, _, lvc = tp.coupling(pc, gc, tst, "", warp, "inverse")
    ((zc * torch.from_numpy(wz)).sum() + (lvc * torch.from_numpy(wl)).sum()).backward()
    pt, gt = torch.from_numpy(p).cuda().requires_grad_(True), torch.from_numpy(g).cuda().requires_grad_(True)
    z, ld, _ = m._engine.run(pt, gt, "inverse", False) if m._engine else (None, None, None)
    if z is None:
        from go_with_the_flows_amd.flows import StackEngine
        m._engine = StackEngine([m]); z, ld, _ = m._engine.run(pt, gt, "inverse", False)
    ((z * torch.from_numpy(wz).cuda()).sum() + (ld * torch.from_numpy(wl).cuda()).sum()).backward()
    print(f"pattern {pi} warp {warp}: fwd {rel(z.detach().cpu(), zc.detach()):.1e} dp {rel(pt.grad.cpu(), pc.grad):.2e} (per-dim " +
          " ".join(f"{rel(pt.grad.cpu()[:, d], pc.grad[:, d]):.1e}" for d in range(3)) + f") dg {rel(gt.grad.cpu(), gc.grad):.2e}")
    for k, prm in m.named_parameters():
        e = rel(prm.grad.cpu(), tst[k].grad)
        if e > 1e-3: print("    BAD", k, f"{e:.2e}", tuple(prm.shape))
    if pi == 4:
        print("hip dp[0,:, :4]\n", pt.grad.cpu()[0, :, :4].numpy()); print("ref\n", pc.grad[0, :, :4].numpy())
        print("wz\n", wz[0, :, :4])
