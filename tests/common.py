"""Shared helpers for the test-suite (golden loading, seeded inputs)."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
import sys  # noqa: E402
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_golden(name):
    """Returns (state_dict of f32 tensors, dict of the other arrays)."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    sd, rest = {}, {}
    for k in z.files:
        if k.startswith("w/"):
            sd[k[2:]] = torch.from_numpy(z[k].astype(np.float32))
        else:
            rest[k] = z[k]
    return sd, rest


def load_into(module, sd, prefix=""):
    """Load the golden weights (keys under ``prefix``) into a vidil_amd module, strictly."""
    own = module.state_dict()
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    missing = [k for k, v in own.items() if v.dtype == torch.float32 and k not in sub]
    assert not missing, f"golden lacks {missing[:5]}"
    msg = module.load_state_dict(sub, strict=False)
    assert all("position_ids" in k for k in msg.missing_keys), msg.missing_keys
    return module


def perturb_(module, seed):
    """Non-trivial LayerNorm gains / biases so a bug in their handling shows."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)


def synthetic_frames(n_videos, frames=8, size=224, first_video=0):
    """BASELINE.md §3 inputs: uint8 [F,S,S,3] from default_rng(1000 + video_idx)."""
    out = np.empty((n_videos, frames, size, size, 3), dtype=np.uint8)
    for v in range(n_videos):
        out[v] = np.random.default_rng(1000 + first_video + v).integers(0, 256, size=(frames, size, size, 3), dtype=np.uint8)
    return out


from oracle.synth_weights import portable_init_, trained_like_  # noqa: E402,F401  (shared with bench.py's parity leg / smoke())


