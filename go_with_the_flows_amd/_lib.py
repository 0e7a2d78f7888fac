"""ctypes binding of libgwtf_hip.so (C ABI: include/gwtf.h).

The product path has no fallback: if the shared library is missing or a call fails this module
raises.  Calls are enqueued on torch's current HIP stream for the tensor's device.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GWTF_LIB') or os.path.join(_HERE, 'libgwtf_hip.so')      # GWTF_LIB: an A/B build (tools/ab_build.sh)
ABI_VERSION = 6

MODE_DIRECT, MODE_INVERSE = 0, 1
STAT_REPLICAS = 64   # GWTF_STAT_REPLICAS in csrc/gwtf_layout.h
_MODES = {'direct': MODE_DIRECT, 'inverse': MODE_INVERSE}

_c_fp = ctypes.c_void_p
_SIGNATURES = {
    'gwtf_abi_version': (ctypes.c_int, []),
    'gwtf_error_string': (ctypes.c_char_p, [ctypes.c_int]),
    'gwtf_diag_stamp': (ctypes.c_int, [_c_fp, _c_fp]),
    'gwtf_padded_width': (ctypes.c_int, [ctypes.c_int]),
    'gwtf_raw_coupling_floats': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'gwtf_packed_w_coupling_floats': (ctypes.c_size_t, [ctypes.c_int]),
    'gwtf_packed_film_coupling_floats': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'gwtf_film_out_floats': (ctypes.c_size_t, [ctypes.c_int]),
    'gwtf_pack_weights': (ctypes.c_int, [_c_fp, _c_fp, _c_fp] + [ctypes.c_int] * 5 + [_c_fp]),
    'gwtf_pack_weights_k': (ctypes.c_int, [_c_fp, _c_fp, _c_fp] + [ctypes.c_int] * 6 + [_c_fp]),
    'gwtf_film_forward': (ctypes.c_int, [_c_fp, _c_fp, _c_fp, _c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_float, ctypes.c_int, _c_fp]),
    'gwtf_film_bn_swish_forward': (ctypes.c_int, [_c_fp] * 3 + [ctypes.c_long] * 3 + [ctypes.c_int] * 3 + [_c_fp] * 5),
    'gwtf_film_bn_swish_backward': (ctypes.c_int, [_c_fp] * 4 + [ctypes.c_long] * 3 + [ctypes.c_int] * 3 + [_c_fp] * 6),
    'gwtf_stack_forward': (ctypes.c_int, [_c_fp] * 8 + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_stack_forward_multi': (ctypes.c_int, [_c_fp] * 8 + [ctypes.POINTER(ctypes.c_int)] + [ctypes.c_int] * 6 +
                                 [ctypes.c_float, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, _c_fp]),
    'gwtf_packed_x_coupling_floats': (ctypes.c_size_t, [ctypes.c_int]),
    'gwtf_pack_weights_exact': (ctypes.c_int, [_c_fp, _c_fp, _c_fp] + [ctypes.c_int] * 5 + [_c_fp]),
    'gwtf_stack_forward_exact': (ctypes.c_int, [_c_fp] * 8 + [ctypes.POINTER(ctypes.c_int)] + [ctypes.c_int] * 6 +
                                 [ctypes.c_float, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_stack_forward_flagging': (ctypes.c_int, [_c_fp] * 8 + [ctypes.POINTER(ctypes.c_int)] + [ctypes.c_int] * 6 +
                                    [ctypes.c_float, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, _c_fp, ctypes.c_int, _c_fp]),
    'gwtf_stack_rerun_flagged': (ctypes.c_int, [_c_fp] * 8 + [ctypes.POINTER(ctypes.c_int)] + [ctypes.c_int] * 6 +
                                 [ctypes.c_float, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, _c_fp, ctypes.c_int, _c_fp]),
    'gwtf_latent_loss_workspace_floats': (ctypes.c_int, [ctypes.c_int] * 2),
    'gwtf_latent_loss_forward': (ctypes.c_int, [_c_fp] * 8 + [ctypes.c_int] * 3 + [ctypes.c_float] * 3 + [_c_fp]),
    'gwtf_latent_loss_backward': (ctypes.c_int, [_c_fp] * 10 + [ctypes.c_int] * 3 + [ctypes.c_float] * 3 + [_c_fp]),
    'gwtf_stack_plan': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)] + [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)]),
    'gwtf_train_moments': (ctypes.c_int, [_c_fp, _c_fp, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_train_fold0': (ctypes.c_int, [_c_fp, _c_fp, ctypes.c_double, ctypes.c_int, _c_fp, _c_fp, _c_fp, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_train_stats': (ctypes.c_int, [_c_fp, _c_fp, _c_fp] + [ctypes.c_int] * 5 + [_c_fp]),
    'gwtf_train_fold1': (ctypes.c_int, [_c_fp, _c_fp, ctypes.c_double, _c_fp, _c_fp, _c_fp] + [ctypes.c_int] * 5 + [_c_fp]),
    'gwtf_train_apply': (ctypes.c_int, [_c_fp] * 10 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_train_forward': (ctypes.c_int, [_c_fp] * 14 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_pack_w1t': (ctypes.c_int, [_c_fp, _c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_train_coupling_backward': (ctypes.c_int, [_c_fp] * 21 + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_int, _c_fp]),
    'gwtf_packed_b_coupling_floats': (ctypes.c_size_t, [ctypes.c_int]),
    'gwtf_pack_folded': (ctypes.c_int, [_c_fp] * 5 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_train_backward': (ctypes.c_int, [_c_fp] * 23 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, _c_fp]),
    'gwtf_coupling_backward': (ctypes.c_int, [_c_fp] * 11 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, _c_fp]),
    'gwtf_coupling_backward_lists': (ctypes.c_int, [_c_fp] * 13 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, _c_fp]),
    'gwtf_stats_backward': (ctypes.c_int, [_c_fp] * 7 + [ctypes.c_int] * 4 + [_c_fp]),
    'gwtf_dw1_partials': (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    'gwtf_dw1_workspace_floats': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'gwtf_dw1_reduce_scratch_floats': (ctypes.c_size_t, [ctypes.c_int]),
    'gwtf_dw1_reduce': (ctypes.c_int, [_c_fp, ctypes.c_int, _c_fp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_mixture_nll': (ctypes.c_int, [_c_fp] * 7 + [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_mixture_nll_backward': (ctypes.c_int, [_c_fp] * 12 + [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_adam_step': (ctypes.c_int, [_c_fp] * 6 + [ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_adam_chunk_elems': (ctypes.c_int, []),
    'gwtf_adam_step_table': (ctypes.c_int, [_c_fp] * 3 + [ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_double,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_nn_distance': (ctypes.c_int, [_c_fp] * 6 + [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_nn_distance_grad': (ctypes.c_int, [_c_fp] * 8 + [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_approx_match': (ctypes.c_int, [_c_fp] * 4 + [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_emd_cost': (ctypes.c_int, [_c_fp] * 4 + [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_match_cost': (ctypes.c_int, [_c_fp] * 4 + [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_match_cost_grad': (ctypes.c_int, [_c_fp] * 5 + [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_encoder_raw_floats': (ctypes.c_size_t, [_c_fp, ctypes.c_int]),
    'gwtf_encoder_packed_floats': (ctypes.c_size_t, [_c_fp, ctypes.c_int]),
    'gwtf_encoder_pack': (ctypes.c_int, [_c_fp, _c_fp, _c_fp, ctypes.c_int, _c_fp]),
    'gwtf_encoder_forward': (ctypes.c_int, [_c_fp] * 4 + [ctypes.c_int, ctypes.c_int, _c_fp, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_supported': (ctypes.c_int, [_c_fp, ctypes.c_int]),
    'gwtf_enc_train_units_floats': (ctypes.c_size_t, [ctypes.c_int]),
    'gwtf_enc_train_act_floats': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'gwtf_enc_train_pack': (ctypes.c_int, [_c_fp] * 3 + [ctypes.c_int, _c_fp]),
    'gwtf_enc_train_pack_all': (ctypes.c_int, [_c_fp] * 10),
    'gwtf_enc_train_xmoments': (ctypes.c_int, [_c_fp, _c_fp, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_fold0': (ctypes.c_int, [_c_fp, ctypes.c_double] + [_c_fp] * 5 + [ctypes.c_float, _c_fp, _c_fp, _c_fp]),
    'gwtf_enc_train_fold': (ctypes.c_int, [_c_fp, ctypes.c_int, ctypes.c_double] + [_c_fp] * 4 + [ctypes.c_float, _c_fp, _c_fp, _c_fp]),
    'gwtf_enc_train_forward': (ctypes.c_int, [ctypes.c_int] + [_c_fp] * 9 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_pool': (ctypes.c_int, [_c_fp] * 6 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_pack_matrix': (ctypes.c_int, [_c_fp, _c_fp, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_top_scatter': (ctypes.c_int, [_c_fp] * 7 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_backward_top': (ctypes.c_int, [_c_fp] * 10 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_top': (ctypes.c_int, [_c_fp] * 7 + [ctypes.c_int, _c_fp]),
    'gwtf_enc_train_bwd_consts': (ctypes.c_int, [_c_fp, ctypes.c_int, ctypes.c_double] + [_c_fp] * 5 + [_c_fp]),
    'gwtf_enc_train_backward': (ctypes.c_int, [ctypes.c_int] + [_c_fp] * 10 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_dw_partial_floats': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'gwtf_enc_train_dw': (ctypes.c_int, [ctypes.c_int] + [_c_fp] * 7 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_dw3': (ctypes.c_int, [_c_fp] * 9 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_stat_compact': (ctypes.c_int, [_c_fp, _c_fp, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_mform_workspace_floats': (ctypes.c_size_t, [ctypes.c_int]),
    'gwtf_enc_train_mform': (ctypes.c_int, [_c_fp] * 5 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_dw3_finish': (ctypes.c_int, [_c_fp] * 6 + [ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_enc_train_dw0_finish': (ctypes.c_int, [_c_fp] * 5 + [ctypes.c_int, _c_fp]),
    'gwtf_prior_raw_floats': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'gwtf_prior_raw_offset': (ctypes.c_size_t, [ctypes.c_int] * 4),
    'gwtf_prior_workspace_floats': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'gwtf_prior_forward': (ctypes.c_int, [_c_fp] * 7 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_prior_backward': (ctypes.c_int, [_c_fp] * 10 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_bn_running_update': (ctypes.c_int, [_c_fp, _c_fp, _c_fp, ctypes.c_int, ctypes.c_int, _c_fp]),
    'gwtf_gather_table': (ctypes.c_int, [_c_fp, _c_fp, ctypes.c_int, _c_fp]),
    'gwtf_mtrain_dw1_floats': (ctypes.c_size_t, [ctypes.c_int] * 3),
    'gwtf_film_heads_slices': (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    'gwtf_film_heads_forward': (ctypes.c_int, [_c_fp] * 7 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, _c_fp]),
    'gwtf_film_heads_backward': (ctypes.c_int, [_c_fp] * 10 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, _c_fp]),
    'gwtf_mtrain_phase': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    'gwtf_mtrain_forward': (ctypes.c_int, [ctypes.c_void_p]),
    'gwtf_mtrain_backward': (ctypes.c_int, [ctypes.c_void_p]),
    'gwtf_mtrain_final_forward_half': (ctypes.c_int, [ctypes.c_int]),
    'gwtf_mtrain_final_backward_half': (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    'gwtf_head_layer_supported': (ctypes.c_int, [ctypes.c_int] * 4),
    'gwtf_head_layer_forward': (ctypes.c_int, [_c_fp] * 8 + [ctypes.c_float, ctypes.c_float] + [ctypes.c_int] * 3 + [_c_fp] * 3 +
                                [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_head_layer_backward': (ctypes.c_int, [_c_fp] * 9 + [ctypes.c_int] * 2 + [_c_fp] * 2 + [ctypes.c_int] + [_c_fp] * 4 +
                                 [ctypes.c_int] * 3 + [_c_fp]),
    'gwtf_head_pair_forward': (ctypes.c_int, [_c_fp] * 9 + [ctypes.c_int] * 4 + [_c_fp]),
    'gwtf_head_pair_backward': (ctypes.c_int, [_c_fp] * 18 + [ctypes.c_int] * 4 + [_c_fp]),
}

PHASE_FWD_INIT, PHASE_FWD_A, PHASE_FWD_B, PHASE_BWD_A, PHASE_BWD_B, PHASE_BWD_C = range(6)


class TrainCtx(ctypes.Structure):
    """GwtfTrainCtx of include/gwtf.h (K-batched, phase-split train pipeline): same field order."""
    _fields_ = ([(n, ctypes.c_int) for n in ('K', 'B', 'N', 'C', 'f', 'G', 'pattern0', 'mode', 'tune')] +
                [('eps', ctypes.c_float), ('n_total', ctypes.c_double)] +
                [(n, ctypes.c_void_p) for n in (
                    'p', 'raw', 'packed_w', 'packed_b', 'film_raw', 'film_rec', 'moments', 'ystats', 'mom_c', 'ys_c', 'bn_batch', 'xbuf',
                    'logdet', 'ps', 'mus', 'logvars', 'g_out', 'g_ld', 'g_ps', 'g_lvs', 'g_bufs', 'g_xa', 'g_xb', 'dw1_ws', 'g_film', 'g_sd0',
                    'g_bias', 'g_stats', 'g_mom', 'g_film_raw', 'g_raw', 'stream')])
EXPORTS = tuple(_SIGNATURES)

# ---- per-call tuning word (include/gwtf.h GWTF_TUNE_*) ------------------------------------------------------------------------
# The library keeps no tuning state: every dispatching entry point takes the word as an argument.  The HOST side keeps the value
# the wrappers pass -- 0 unless a test / calibration tool changes it inside `with tuning(...)` (restored on exit, also on an
# exception).
TUNE_GENERIC_BODY, TUNE_SMALL_LIGHT_TILE, TUNE_SINGLE_TILE = 1 << 30, 1 << 29, 1 << 28
_TUNE = [0]


def tune_word():
    return _TUNE[0]


def set_tuning(word=0):
    """Set the tuning word the wrappers pass from now on (tests: an autouse fixture resets it); prefer `with tuning(...)`."""
    _TUNE[0] = int(word)


class tuning:
    """with tuning(points_per_wave=64, generic_body=True): ...   -- tile size / coupling body forced for the calls inside."""

    def __init__(self, points_per_wave=0, generic_body=False, small_light_tile=False, word=None):
        self.word = (int(points_per_wave) & 0xffff) | (TUNE_GENERIC_BODY if generic_body else 0) | \
            (TUNE_SMALL_LIGHT_TILE if small_light_tile else 0) if word is None else int(word)

    def __enter__(self):
        self.saved, _TUNE[0] = _TUNE[0], self.word
        return self

    def __exit__(self, *exc):
        _TUNE[0] = self.saved
        return False

_lib = None


class GwtfError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises GwtfError when the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GwtfError(f'{LIB_PATH} not found: build it with `make -C go_with_the_flows_amd/csrc` '
                            '(or `python -c "import __graft_entry__ as g; g.build()"`). There is no fallback path.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.gwtf_abi_version() != ABI_VERSION:
            raise GwtfError(f'libgwtf_hip.so ABI {handle.gwtf_abi_version()} != expected {ABI_VERSION}; rebuild')
        _lib = handle
    return _lib


def check(code):
    if code != 0:
        raise GwtfError(f'libgwtf_hip call failed ({code}): {lib().gwtf_error_string(code).decode()}')


def _ptr(t, name):
    """Device pointer of a tensor after the input checks the reference's native ops apply
    (is-device + contiguous: lib/metrics/pytorch_structural_losses/src/structural_loss.cpp:10-12)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GwtfError(f'{name} must live on a HIP device (got {t.device}); there is no CPU path')
    if not t.is_contiguous():
        raise GwtfError(f'{name} must be contiguous')
    if t.dtype != torch.float32:
        raise GwtfError(f'{name} must be float32 (got {t.dtype})')
    return t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def padded_width(f):
    return lib().gwtf_padded_width(f)


def pack_weights(raw, C, f, G, training, pattern0=0, K=1, stack_only=False):
    """Packed stack / FiLM weights of K concatenated stacks of C couplings each (raw: K*C coupling records).  stack_only (train
    pipeline): the stack weights alone -- its FiLM heads read the raw arena in place; returns (pw, None)."""
    L = lib()
    pw = torch.empty(K * C * L.gwtf_packed_w_coupling_floats(f), device=raw.device, dtype=torch.float32)
    if stack_only:
        if raw.numel() != K * C * L.gwtf_raw_coupling_floats(f, G):
            raise GwtfError(f'raw arena has {raw.numel()} floats, expected {K * C * L.gwtf_raw_coupling_floats(f, G)}')
        with torch.cuda.device(raw.device):
            check(L.gwtf_pack_weights_k(_ptr(raw, 'raw'), _ptr(pw, 'packed_w'), None, K, C, f, G, int(pattern0), 2, _stream(raw)))
        return pw, None
    pf = torch.empty(K * C * L.gwtf_packed_film_coupling_floats(f, G), device=raw.device, dtype=torch.float32)
    if raw.numel() != K * C * L.gwtf_raw_coupling_floats(f, G):
        raise GwtfError(f'raw arena has {raw.numel()} floats, expected {K * C * L.gwtf_raw_coupling_floats(f, G)}')
    with torch.cuda.device(raw.device):
        check(L.gwtf_pack_weights_k(_ptr(raw, 'raw'), _ptr(pw, 'packed_w'), _ptr(pf, 'packed_film'), K, C, f, G, int(pattern0),
                                    int(bool(training)), _stream(raw)))
    return pw, pf


def film_forward(g, packed_film, C, f, eps, training, want_stats=False):
    """eval: the (B,C,6FP+4) record the stack kernel consumes.  training: RAW FiLM {a,b} as (B,C,2,2,FP)."""
    L = lib()
    B, G = g.shape
    if training:
        out = torch.empty(B, C, 2, 2, L.gwtf_padded_width(f), device=g.device, dtype=torch.float32)
    else:
        out = torch.empty(B, C, L.gwtf_film_out_floats(f), device=g.device, dtype=torch.float32)
    stats = torch.empty(C, 2, 2, 2, f, device=g.device, dtype=torch.float32) if (training and want_stats) else None
    with torch.cuda.device(g.device):
        check(L.gwtf_film_forward(_ptr(g, 'g'), _ptr(packed_film, 'packed_film'), _ptr(out, 'film_out'),
                                  _ptr(stats, 'bn_stats'), B, G, C, f, float(eps), int(bool(training)), _stream(g)))
    return (out, stats) if want_stats else out


# ---- the exact-fp32 contraction body (csrc/gwtf_stack_exact.hip) ---------------------------------------------------------------
# EXACT[0]: run the stack on the fp32 matrix instruction instead of the split-f16 one (tests, bench.py's comparison point;
# `with exact_fp32():`).  Otherwise, when the caller hands the exact record (packed_x) along, every split launch is followed by the
# RE-RUN launch: tiles holding a point the split kernel flagged out of range (NaN) are recomputed exactly, all others exit at once.
EXACT = [False]


class exact_fp32:
    def __enter__(self):
        self._old, EXACT[0] = EXACT[0], True
        return self

    def __exit__(self, *exc):
        EXACT[0] = self._old
        return False


def pack_weights_exact(raw, packed_film, C, f, G, pattern0=0, K=1):
    """GwtfPackX records of K concatenated stacks of C couplings (the fp32 operands of the exact contraction body)."""
    L = lib()
    n = K * C * L.gwtf_packed_x_coupling_floats(f)
    # behind the records: the work list of the flagging launch / re-run pair (include/gwtf.h), zero once -- the pair keeps it zero.
    # It lives and dies with the record it serves; like the record it belongs to one stream at a time.
    px = torch.empty(n + WORKLIST_INTS, device=raw.device, dtype=torch.float32)
    px[n:].zero_()
    with torch.cuda.device(raw.device):
        check(L.gwtf_pack_weights_exact(_ptr(raw, 'raw'), _ptr(packed_film, 'packed_film'), _ptr(px, 'packed_x'), K, C, f, G,
                                        int(pattern0), _stream(raw)))
    return px


WORKLIST_INTS = 2 + 2 * 2048          # GWTF_WORKLIST_INTS of include/gwtf.h


def _worklist_ptr(packed_x, K, C, f):
    """Device address of the work list behind the K * C exact records of `packed_x` (pack_weights_exact), or None."""
    n = K * C * lib().gwtf_packed_x_coupling_floats(f)
    if packed_x is None or packed_x.numel() != n + WORKLIST_INTS or os.environ.get('GWTF_NO_RERUN_WORKLIST') == '1':
        return None
    return packed_x.data_ptr() + 4 * n


def _rerun_launch(p, packed_x, film, out, logdet, lp, seg, K, C, f, pattern0, eps, mode, p_stride, out_stride, wl):
    """The exact-fp32 re-run of the tiles the preceding split launch flagged: from its work list, or by looking at every tile."""
    if wl is None:
        return _exact_launch(p, packed_x, film, out, logdet, lp, seg, K, C, f, pattern0, eps, mode, p_stride, out_stride, 1)
    B, _, N = p.shape
    check(lib().gwtf_stack_rerun_flagged(_ptr(p, 'p'), _ptr(packed_x, 'packed_x'), _ptr(film, 'film'), _ptr(out, 'out'),
                                         _ptr(logdet, 'logdet'), lp[0], lp[1], lp[2], seg, K, B, N, C, f, pattern0, float(eps),
                                         _MODES[mode], p_stride, out_stride, wl, _TUNE[0] & 0xffff, _stream(p)))


def _exact_launch(p, packed_x, film, out, logdet, lp, seg, K, C, f, pattern0, eps, mode, p_stride, out_stride, only_flagged):
    L = lib()
    B, _, N = p.shape
    check(L.gwtf_stack_forward_exact(_ptr(p, 'p'), _ptr(packed_x, 'packed_x'), _ptr(film, 'film'), _ptr(out, 'out'),
                                     _ptr(logdet, 'logdet'), lp[0], lp[1], lp[2], seg, K, B, N, C, f, pattern0, float(eps),
                                     _MODES[mode], p_stride, out_stride, int(only_flagged), _TUNE[0] & 0xffff, _stream(p)))


def stack_forward(p, packed_w, film, C, f, pattern0, eps, mode, want_lists, packed_x=None):
    L = lib()
    B, three, N = p.shape
    if three != 3:
        raise GwtfError(f'p must be (B,3,N), got {tuple(p.shape)}')
    if film.shape[0] != B or film.shape[1] != C:
        raise GwtfError(f'film is {tuple(film.shape)}, expected ({B},{C},...)')
    out = torch.empty_like(p)
    logdet = torch.empty_like(p)
    lists = torch.empty(3, C, B, 3, N, device=p.device, dtype=torch.float32) if want_lists else None
    lp = [lists[i].data_ptr() for i in range(3)] if want_lists else [None, None, None]
    with torch.cuda.device(p.device):
        if EXACT[0]:
            if packed_x is None:
                raise GwtfError('exact_fp32: this call site has no exact record (packed_x)')
            _exact_launch(p, packed_x, film, out, logdet, lp, None, 1, C, f, pattern0, eps, mode, 0, 0, 0)
            return out, logdet, lists
        wl = _worklist_ptr(packed_x, 1, C, f)
        if wl is None:
            check(L.gwtf_stack_forward(_ptr(p, 'p'), _ptr(packed_w, 'packed_w'), _ptr(film, 'film'), _ptr(out, 'out'),
                                       _ptr(logdet, 'logdet'), lp[0], lp[1], lp[2], B, N, C, f, pattern0, float(eps),
                                       _MODES[mode], _TUNE[0], _stream(p)))
        else:
            check(L.gwtf_stack_forward_flagging(_ptr(p, 'p'), _ptr(packed_w, 'packed_w'), _ptr(film, 'film'), _ptr(out, 'out'),
                                                _ptr(logdet, 'logdet'), lp[0], lp[1], lp[2], None, 1, B, N, C, f, pattern0,
                                                float(eps), _MODES[mode], 0, 0, wl, _TUNE[0], _stream(p)))
        if packed_x is not None:
            _rerun_launch(p, packed_x, film, out, logdet, lp, None, 1, C, f, pattern0, eps, mode, 0, 0, wl)
    return out, logdet, lists


def stack_forward_multi(p, packed_w, film, K, C, f, pattern0, eps, mode, segments=None, shared_points=True, out=None, logdet=None,
                        packed_x=None):
    """K components in one launch.  shared_points=True: every component maps all of p -> outputs (K,B,3,N).
    Otherwise ``segments`` (list of K (begin,end)) partitions the N points among the components -> (B,3,N).
    out / logdet: optional preallocated result tensors (a timing probe brackets the launch alone with them)."""
    L = lib()
    B, three, N = p.shape
    if three != 3:
        raise GwtfError(f'p must be (B,3,N), got {tuple(p.shape)}')
    if film.shape[0] != B or film.shape[1] != K * C:
        raise GwtfError(f'film is {tuple(film.shape)}, expected ({B},{K * C},...)')
    if (out is None) != (logdet is None):
        raise GwtfError('stack_forward_multi: pass out and logdet together (both preallocated) or neither')
    seg = None
    if segments is not None:
        flat = [int(v) for be in segments for v in be]
        if len(flat) != 2 * K:
            raise GwtfError('segments must hold K (begin, end) pairs')
        seg = (ctypes.c_int * (2 * K))(*flat)
    if shared_points:
        if out is None:
            out = torch.empty(K, B, 3, N, device=p.device, dtype=torch.float32)
            logdet = torch.empty(K, B, 3, N, device=p.device, dtype=torch.float32)
        stride = B * 3 * N
    else:
        if out is None:
            # points outside every segment are not touched by the kernel: define them -- unless the segments tile [0, N) (the
            # sampling partition of flow_mixture.py:146-177 always does): two fill launches less per call
            covered = sorted((int(b), int(e)) for b, e in segments if int(e) > int(b))
            tiled = bool(covered) and covered[0][0] == 0 and covered[-1][1] == N and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
            make = torch.empty if tiled else torch.zeros
            out = make(B, 3, N, device=p.device, dtype=torch.float32)
            logdet = make(B, 3, N, device=p.device, dtype=torch.float32)
        stride = 0
    with torch.cuda.device(p.device):
        if EXACT[0]:
            if packed_x is None:
                raise GwtfError('exact_fp32: this call site has no exact record (packed_x)')
            _exact_launch(p, packed_x, film, out, logdet, [None] * 3, seg, K, C, f, pattern0, eps, mode, 0, stride, 0)
            return out, logdet
        wl = _worklist_ptr(packed_x, K, C, f)
        if wl is None:
            check(L.gwtf_stack_forward_multi(_ptr(p, 'p'), _ptr(packed_w, 'packed_w'), _ptr(film, 'film'), _ptr(out, 'out'),
                                             _ptr(logdet, 'logdet'), None, None, None, seg, K, B, N, C, f, pattern0,
                                             float(eps), _MODES[mode], 0, stride, _TUNE[0], _stream(p)))
        else:
            check(L.gwtf_stack_forward_flagging(_ptr(p, 'p'), _ptr(packed_w, 'packed_w'), _ptr(film, 'film'), _ptr(out, 'out'),
                                                _ptr(logdet, 'logdet'), None, None, None, seg, K, B, N, C, f, pattern0,
                                                float(eps), _MODES[mode], 0, stride, wl, _TUNE[0], _stream(p)))
        if packed_x is not None:
            _rerun_launch(p, packed_x, film, out, logdet, [None] * 3, seg, K, C, f, pattern0, eps, mode, 0, stride, wl)
    return out, logdet


def dw1_workspace(f, B, N, device, passes=1):
    """Workspace for the per-workgroup dW1 partials of `passes` backward passes over B x N points (csrc/gwtf_bwd.hip)."""
    L = lib()
    return torch.empty(passes * L.gwtf_dw1_workspace_floats(f, B, N) + L.gwtf_dw1_reduce_scratch_floats(f), device=device,
                       dtype=torch.float32)


def dw1_reduce(ws, passes, f, B, N, out=None, branch_stride=None):
    """Sum the partials -> (2,f,f) sd1 weight gradient (deterministic).  With ``branch_stride`` the two (f,f) blocks go to
    out + br*branch_stride (in place into a gradient record)."""
    if out is None:
        out = torch.empty(2, f, f, device=ws.device, dtype=torch.float32)
    check(lib().gwtf_dw1_reduce(_ptr(ws, 'dw1_ws'), passes, out.data_ptr(), f * f if branch_stride is None else branch_stride,
                                f, B, N, _stream(ws)))
    return out


def mixture_nll(z, logdet, mu0, lv0, logits, want_point_lse=False):
    L = lib()
    K, B, _, N = z.shape
    nll = torch.empty(B, device=z.device, dtype=torch.float32)
    plse = torch.empty(B, N, device=z.device, dtype=torch.float32) if want_point_lse else None
    with torch.cuda.device(z.device):
        check(L.gwtf_mixture_nll(_ptr(z, 'z'), _ptr(logdet, 'logdet'), _ptr(mu0, 'mu0'), _ptr(lv0, 'lv0'),
                                 _ptr(logits, 'logits'), _ptr(plse, 'point_lse'), _ptr(nll, 'nll_shape'), K, B, N,
                                 _stream(z)))
    return (nll, plse) if want_point_lse else nll


def train_forward(p, g, raw, C, f, G, pattern0, eps, mode, want_lists):
    """Train-mode (batch-statistic BatchNorm) forward of one coupling stack on one rank, entirely in HIP: the FiLM heads with
    batch statistics over the B latent rows (gwtf_film_forward, training=1), then fold0 -> stats -> fold1 -> apply per
    coupling from ONE C call (csrc/gwtf_train.hip).  Data-parallel runs take autograd.train_density_forward_multi instead.
    Returns out, logdet, lists, bn_batch (C,2,4,2,f) = {batch mean, unbiased batch var} of the 8 BatchNorms per
    coupling (kind 0 sd0_bn, 1 sd1_bn, 2 film_w0_bn, 3 film_b0_bn; branch 0 logvar, 1 mu)."""
    L = lib()
    B, _, N = p.shape
    dev = p.device
    FP = L.gwtf_padded_width(f)
    FS = L.gwtf_film_out_floats(f)
    st = _stream(p)
    if B < 2:
        raise ValueError('train-mode BatchNorm needs more than 1 shape per batch (torch raises the same)')
    with torch.cuda.device(dev):
        pw, pf = pack_weights(raw, C, f, G, True, pattern0)
        film_raw, fstats = film_forward(g, pf, C, f, eps, True, want_stats=True)
        # a non-finite parameter anywhere in a branch record -> NaN FiLM scale -> NaN outputs (the kernels' v_max ReLU alone
        # would turn e.g. a NaN sd0 weight into a zero activation; reference training.py:43-46 aborts on a NaN loss)
        film_raw[:, :, :, 0] += (raw.view(C, 2, -1).sum(-1) * 0.0).view(1, C, 2, 1)
        mom = torch.zeros(C + 1, STAT_REPLICAS * 16, device=dev, dtype=torch.float32)
        ystats = torch.zeros(C, STAT_REPLICAS * 2 * FP * 2, device=dev, dtype=torch.float32)
        bn_batch = torch.zeros(C, 2, 4, 2, f, device=dev, dtype=torch.float32)
        film_rec = torch.empty(B, C, FS, device=dev, dtype=torch.float32)
        xbuf = torch.empty(2, B, 3, N, device=dev, dtype=torch.float32)
        logdet = torch.empty_like(p)
        lists = torch.empty(3, C, B, 3, N, device=dev, dtype=torch.float32) if want_lists else None
        lp = [lists[i].data_ptr() for i in range(3)] if want_lists else [None, None, None]
        check(L.gwtf_train_forward(_ptr(p, 'p'), _ptr(raw, 'raw'), pw.data_ptr(), None, film_raw.data_ptr(), mom.data_ptr(),
                                   ystats.data_ptr(), bn_batch.data_ptr(), film_rec.data_ptr(), xbuf.data_ptr(),
                                   logdet.data_ptr(), lp[0], lp[1], lp[2], B, N, C, f, G, pattern0, float(eps),
                                   _MODES[mode], _TUNE[0], st))
        out = xbuf[(C - 1) & 1]
        # per-shape FiLM BatchNorms: biased batch var -> unbiased
        bn_batch[:, :, 2:4, 0, :] = fstats[:, :, :, 0, :]
        bn_batch[:, :, 2:4, 1, :] = fstats[:, :, :, 1, :] * (B / (B - 1.0))
    return out, logdet, lists, bn_batch


def stack_plan(K, B, N, f, segments=None, word=None):
    """The tile plan gwtf_stack_forward* would use (no launch): (points per wave, workgroups) -- host-only, works without a GPU."""
    seg = None
    if segments is not None:
        flat = [int(v) for be in segments for v in be]
        seg = (ctypes.c_int * len(flat))(*flat)
    out = (ctypes.c_int * 4)()
    check(lib().gwtf_stack_plan(seg, K, B, N, f, _TUNE[0] if word is None else int(word), out))
    return out[0], out[1]
