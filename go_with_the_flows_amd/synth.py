"""Deterministic synthetic parameters and inputs (bench, smoke test, golden fixtures).

The reference initialises its last layers at N(0, 0.01) so a fresh decoder is almost the identity
and its log-dets are ~1e-2: useless for parity tests and unrepresentative for timing.  ``synth_state``
fills a ``state_dict`` (keys/shapes taken from the module it is given) with values that make every
BatchNorm, FiLM head and output layer matter: O(1) log-dets, non-trivial running statistics.
Generation is pure numpy from a seed, so fixtures store the seed instead of megabytes of weights.
"""
import numpy as np
import torch


CONDITIONED_GAIN = 0.3   # output_gain of the well-conditioned state: a deep stack then maps 0.3 N(0,1) clouds to |z| of a few units


def synth_state(template_state, seed, output_gain=1.0):
    """template_state: mapping key -> tensor/array (only shapes/dtypes are used).  Returns numpy dict.
    output_gain scales every coupling's output layer (sd2 weight and bias) AFTER the draw, so the random stream -- and every
    golden fixture generated with the default -- is unchanged.  The default (1.0) is the bench / fixture state: O(1) log-dets per
    coupling, which a 33-coupling inverse pass compounds to |z| ~ 5e3 on some points (fp32 evaluation is then ill-conditioned:
    the stated ABSOLUTE tolerance is below fp32's own noise there).  CONDITIONED_GAIN gives what a trained model does -- clouds
    mapped to a few units -- and is the state on which the absolute tolerance is asserted without reference to fp32 noise
    (tests/test_gpu_fullgrid.py)."""
    rng = np.random.default_rng(seed)
    out = {}
    for key in template_state:  # insertion order of the module's state_dict: deterministic
        shape = tuple(template_state[key].shape)
        leaf = ('', '') + tuple(key.rsplit('.', 2))
        name, kind = leaf[-2], leaf[-1]
        if kind in ('g0_prior_mus', 'g0_prior_logvars', 'mixture_weights_logits'):      # top-level model parameters
            v = rng.normal(0.0, 0.3, shape)
        elif kind == 'p_prior_mus':
            v = np.zeros(shape)
        elif kind == 'p_prior_logvar':
            v = np.full(shape, -1.0)
        elif kind == 'eps':
            v = np.full(shape, 1e-6, np.float32)
        elif kind == 'num_batches_tracked':
            v = np.zeros(shape, np.int64)
        elif name.endswith('_bn'):
            if kind == 'weight':
                v = rng.uniform(0.6, 1.4, shape)
            elif kind == 'bias':
                v = rng.normal(0.0, 0.1, shape)
            elif kind == 'running_mean':
                v = rng.normal(0.0, 0.15, shape)
            else:  # running_var
                v = rng.uniform(0.5, 1.5, shape)
        elif name.endswith('sd0') or name.endswith('sd1') or name == 'init_sd' or (name.startswith('mlp') and kind == 'weight'):
            fan_in = shape[-1]
            v = rng.uniform(-1.0, 1.0, shape) * np.sqrt(3.0 / fan_in)
        elif name.endswith('sd2'):
            v = (rng.normal(0.0, 0.15, shape) if kind == 'weight' else rng.normal(0.0, 0.1, shape)) * output_gain
        elif name.endswith('film_w0') or name.endswith('film_b0'):
            v = rng.uniform(-1.0, 1.0, shape) / np.sqrt(shape[-1])
        elif name.endswith('film_w1') or name.endswith('film_b1') or name in ('mu_mlp0', 'logvar_mlp0', 'mu_mlp1', 'logvar_mlp1'):
            v = rng.normal(0.0, 0.08, shape) if kind == 'weight' else rng.normal(0.0, 0.1, shape)
        else:
            raise KeyError(f'synth_state: unknown parameter kind {key}')
        out[key] = np.asarray(v, dtype=np.int64 if kind == 'num_batches_tracked' else np.float32)
    return out


def load_synth_(module, seed, output_gain=1.0):
    """Overwrite ``module``'s parameters/buffers in place with ``synth_state``; returns the numpy state."""
    st = synth_state(module.state_dict(), seed, output_gain)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return st


def synth_inputs(B, N, G, seed):
    """Clouds ~ 0.3*N(0,1) (ShapeNet clouds are unit-normalised then halved) and latents ~ N(0,1)."""
    rng = np.random.default_rng(seed)
    p = (0.3 * rng.standard_normal((B, 3, N))).astype(np.float32)
    g = rng.standard_normal((B, G)).astype(np.float32)
    return p, g
