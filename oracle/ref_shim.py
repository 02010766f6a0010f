"""oracle/ref_shim.py — TEST INFRASTRUCTURE (only usable where /root/reference exists).

Imports the reference's own ``models/vit.py`` and ``models/med.py`` in THIS
container so the CPU restatement in ``oracle/`` can be pinned against them and
golden vectors can be generated (tests/golden/make_golden.py).  Nothing in the
product, in ``-m gpu`` tests, in ``smoke()`` or in ``bench.py`` imports this: the
reference tree does not exist on the GPU box.

The reference needs ``timm`` and ``fairscale`` (absent here) and three helpers
that moved inside ``transformers`` since the 4.15 the reference targets
(models/med.py:39-44).  The stubs below carry no arithmetic of their own except
``PatchEmbed`` = Conv2d(kernel=stride=patch) + flatten + transpose, which is what
timm's class does (models/vit.py:144-145 relies on it).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VIDIL_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "med.py"))


def _install_stubs():
    import torch
    import torch.nn as nn
    import transformers  # noqa: F401  (must be imported before the aliases below)
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    for name in ("apply_chunking_to_forward", "find_pruneable_heads_and_indices", "prune_linear_layer"):
        if not hasattr(mu, name) and hasattr(pu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            self.img_size = (img_size, img_size)
            self.patch_size = (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    class DropPath(nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("timm")
    mod("timm.models")
    mod("timm.models.vision_transformer", _cfg=lambda **k: {}, PatchEmbed=PatchEmbed)
    mod("timm.models.registry", register_model=lambda f: f)
    mod("timm.models.layers", trunc_normal_=trunc_normal_, DropPath=DropPath)
    mod("timm.models.helpers", named_apply=None, adapt_input_conv=None)
    mod("timm.models.hub", download_cached_file=None)
    mod("fairscale")
    mod("fairscale.nn")
    mod("fairscale.nn.checkpoint")
    mod("fairscale.nn.checkpoint.checkpoint_activations", checkpoint_wrapper=lambda m, **k: m)


_loaded = None


def load():
    """Return (vit_module, med_module) = the reference's models.vit / models.med."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_stubs()
    import importlib.util

    def _load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    vit = _load("_vidil_ref_vit", "models/vit.py")
    med = _load("_vidil_ref_med", "models/med.py")
    # transformers >= 5 removed two things models/med.py:591,769 call on PreTrainedModel
    pm = med.BertPreTrainedModel
    pm.init_weights = lambda self: self.apply(self._init_weights)
    if not hasattr(pm, "get_head_mask"):
        pm.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    _loaded = (vit, med)
    return _loaded


def med_config(encoder_width=768):
    """BertConfig built from the reference's configs/med_config.json."""
    _, med = load()
    cfg = med.BertConfig.from_json_file(os.path.join(REFERENCE_ROOT, "configs", "med_config.json"))
    cfg.encoder_width = encoder_width
    return cfg
