"""Shim for ``from models.blip_itm import ...`` -> vidil_amd.blip_itm (the HIP-backed ITM filter)."""
from vidil_amd.blip_itm import BLIP_ITM, blip_itm  # noqa: F401
