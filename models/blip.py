"""Shim for ``from models.blip import ...`` -> vidil_amd.blip (the HIP-backed BLIP captioner)."""
from vidil_amd.blip import (BLIP_Decoder, blip_decoder, create_vit, init_tokenizer, is_url,  # noqa: F401
                            load_checkpoint)
