"""Shared test helpers: rebuild the seeded weights the fixtures were generated with."""
import numpy as np
import torch

import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import synth_state


def decoder_and_state(L, f, G, seed):
    m = gw.LocalCondRNVPDecoder(L, f, G)
    st = synth_state(m.state_dict(), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return m, st


def coupling_and_state(f, G, warp, seed):
    m = gw.CondRealNVPFlow3D(f, G, warp_inds=list(warp))
    st = synth_state(m.state_dict(), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return m, st


def triple_and_state(f, G, pattern, seed):
    m = gw.CondRealNVPFlow3DTriple(f, G, pattern=pattern)
    st = synth_state(m.state_dict(), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return m, st


def state64(st):
    return {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in st.items()}


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
