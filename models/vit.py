"""Shim for ``from models.vit import ...`` -> vidil_amd.vit."""
from vidil_amd.vit import VisionTransformer, interpolate_pos_embed  # noqa: F401
