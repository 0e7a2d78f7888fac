"""The dense algebra between the encoder's training kernels (csrc/gwtf_encoder_glue.hip, through the C ABI) against float64 torch:
replica sums, the top layer's M form with its power-of-two operand scale and fragment images, the dW3 / dW0 finishing sums.
(The encoder's gradient goldens in test_gpu_encoder.py cover them end to end; these pin each entry point on its own.)  Needs an MI355X."""
import numpy as np
import pytest
import torch

from go_with_the_flows_amd import _lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rnd(seed, *shape, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


def stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize('R,n', [(64, 768), (64, 1), (64, 1024 + 5), (7, 130), (1, 64), (3, 1)])
def test_stat_compact_is_the_replica_sum(R, n):
    slab = rnd(R * 1000 + n, R, n)
    out = torch.full((n,), float('nan'), device=DEV)
    _lib.check(_lib.lib().gwtf_stat_compact(slab.data_ptr(), out.data_ptr(), R, n, stream()))
    want = slab.double().sum(0)
    assert float((out.double() - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))
    again = torch.empty_like(out)
    _lib.check(_lib.lib().gwtf_stat_compact(slab.data_ptr(), again.data_ptr(), R, n, stream()))
    assert torch.equal(out, again)                       # fixed order: the same bits on every run


def unpack_units(units, rows, kdim):
    """Inverse of the fragment order of enc_train_pack_kernel (gwtf_encoder_train.hip:69): -> hi + lo as a (rows, kdim) float64 matrix."""
    L = _lib.lib()
    KS = kdim // 32
    u = units.view(torch.float16).view(rows // 16 * KS, 2, 64, 8).double().cpu().numpy()       # [unit][part][lane][e]
    M = np.zeros((rows, kdim))
    for unit in range(u.shape[0]):
        m, ks = divmod(unit, KS)
        for lane in range(64):
            row, q = 16 * m + (lane & 15), lane >> 4
            for e in range(8):
                k = 32 * ks + 16 * (e >> 2) + 4 * q + (e & 3)
                M[row, k] = u[unit, 0, lane, e] + u[unit, 1, lane, e]
    return M


@pytest.mark.parametrize('scale', [1.0, 3e-7, 2.0 ** -3, 4096.0])
def test_mform_matrix_scale_images_and_vector(scale):
    L = _lib.lib()
    C3, C4 = 256, 512
    W3 = rnd(1, C4, C3, scale=0.08)
    bconst = torch.cat([rnd(2, C4), rnd(3, C4) * scale, rnd(4, C4), torch.zeros(4, device=DEV)]).contiguous()      # s | Q | R | {..}
    units = torch.empty(L.gwtf_enc_train_units_floats(3) // 2, device=DEV)
    mconst = torch.full((C3 + 4,), float('nan'), device=DEV)
    ws = torch.empty(L.gwtf_enc_train_mform_workspace_floats(C3), device=DEV)
    _lib.check(L.gwtf_enc_train_mform(W3.data_ptr(), bconst.data_ptr(), ws.data_ptr(), units.data_ptr(), mconst.data_ptr(), C3, C4, stream()))
    q, r = bconst[C4:2 * C4].double(), bconst[2 * C4:3 * C4].double()
    M = (W3.double() * q[:, None]).t() @ W3.double()
    k = 8.0 - np.floor(np.log2(float(M.abs().max())))
    got_M = ws[:C3 * C3].view(C3, C3).double()
    assert float((got_M - M).abs().max()) <= 2e-6 * float(M.abs().max())
    assert float(mconst[C3]) == 2.0 ** -k and torch.count_nonzero(mconst[C3 + 1:]) == 0
    v = W3.double().t() @ r
    assert float((mconst[:C3].double() - v).abs().max()) <= 2e-6 * max(1.0, float(v.abs().max()))
    # the images hold M 2^k as hi + lo f16 pairs: 22 bits of each element, largest magnitude in [2^8, 2^9)
    img = unpack_units(units, C3, C3)
    want = got_M.cpu().numpy() * 2.0 ** k
    assert 256.0 <= np.abs(want).max() < 512.0
    assert np.abs(img - want).max() <= 2.0 ** -11 * 2.0 ** -10 * 512.0 * 1.01 + 2.0 ** -24        # lo's rounding of a value < 512 (+ the f16 subnormal step)


def test_dw3_and_dw0_finishing_sums():
    L = _lib.lib()
    C1, C3, C4 = 64, 256, 512
    b3 = torch.cat([rnd(10, 3 * C4), torch.zeros(4, device=DEV)]).contiguous()
    S, W3, gram, a2sum = rnd(11, C4, C3), rnd(12, C4, C3, scale=0.1), rnd(13, C3, C3), rnd(14, C3)
    dW3 = torch.empty(C4, C3, device=DEV)
    _lib.check(L.gwtf_enc_train_dw3_finish(b3.data_ptr(), S.data_ptr(), W3.data_ptr(), gram.data_ptr(), a2sum.data_ptr(), dW3.data_ptr(),
                                           C3, C4, stream()))
    s, q, r = (b3[i * C4:(i + 1) * C4].double() for i in range(3))
    want = s[:, None] * S.double() + q[:, None] * (W3.double() @ gram.double()) + r[:, None] * a2sum.double()[None, :]
    assert float((dW3.double() - want).abs().max()) <= 3e-6 * float(want.abs().max())

    b0 = torch.cat([rnd(20, 3 * C1), torch.zeros(4, device=DEV)]).contiguous()
    red5, W0, m = rnd(21, 5, C1), rnd(22, C1, 3), rnd(23, 12)
    dW0 = torch.empty(C1, 3, device=DEV)
    _lib.check(L.gwtf_enc_train_dw0_finish(b0.data_ptr(), red5.data_ptr(), W0.data_ptr(), m.data_ptr(), dW0.data_ptr(), C1, stream()))
    s0, q0, r0 = (b0[i * C1:(i + 1) * C1].double() for i in range(3))
    md = m.double()
    mxx = torch.stack([md[3], md[4], md[5], md[4], md[6], md[7], md[5], md[7], md[8]]).view(3, 3)
    want0 = s0[:, None] * red5.double()[2:5].t() + q0[:, None] * (W0.double() @ mxx) + r0[:, None] * md[None, :3]
    assert float((dW0.double() - want0).abs().max()) <= 3e-6 * float(want0.abs().max())


def test_bad_arguments_are_refused():
    L = _lib.lib()
    x = torch.zeros(64, device=DEV)
    assert L.gwtf_stat_compact(None, x.data_ptr(), 4, 16, stream()) != 0
    assert L.gwtf_stat_compact(x.data_ptr(), x.data_ptr(), 0, 16, stream()) != 0
    assert L.gwtf_enc_train_mform(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 250, 512, stream()) != 0
    assert L.gwtf_enc_train_dw3_finish(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 100, 512, stream()) != 0
    assert L.gwtf_enc_train_dw0_finish(x.data_ptr(), None, x.data_ptr(), x.data_ptr(), x.data_ptr(), 64, stream()) != 0
