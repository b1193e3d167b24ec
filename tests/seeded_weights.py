"""Deterministic weights for the consumer-side goldens (G18): the generator (tests/golden/make_goldens.py, which loads them into
the REFERENCE's modules) and the GPU tests (which load them into v2v_amd's modules under the same state_dict keys) both call
`seeded_state` -- so the 10.7 M parameters of the recurrent UNet never have to be stored, only their recipe.

NumPy's Generator streams are stable across versions by policy (NEP 19); each tensor has its own stream keyed by
(seed, position in the key order), values uniform in +-bound with bound = gain / sqrt(fan_in) (the shape of PyTorch's default
Conv2d initialisation; biases use the weight's fan_in)."""
import numpy as np


def seeded_state(shapes, seed, gain=1.0):
    """shapes: ordered {state_dict key: shape}.  Returns {key: float32 ndarray}."""
    out, fan_in = {}, 1
    for i, (key, shape) in enumerate(shapes.items()):
        shape = tuple(int(s) for s in shape)
        if key.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
        bound = gain / np.sqrt(fan_in)
        g = np.random.Generator(np.random.PCG64([int(seed), i]))
        out[key] = g.uniform(-bound, bound, size=shape).astype(np.float32)
    return out


def seeded_input(seed, *shape):
    """float32 N(0,1) tensor values for a test input: same recipe in the generator and in the tests."""
    return np.random.Generator(np.random.PCG64(int(seed))).standard_normal(tuple(int(s) for s in shape)).astype(np.float32)


def load_seeded(module, seed, gain=1.0):
    """Fill `module` (any torch.nn.Module) with seeded_state values for its own state_dict keys and shapes; returns the dict."""
    import torch
    sd = module.state_dict()
    vals = seeded_state({k: tuple(v.shape) for k, v in sd.items()}, seed, gain)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    return vals
