"""Import shim: ``from models.blip_retrieval import blip_retrieval`` (run_visual_tokenization.py:18)."""
from vidil_amd.blip_retrieval import BLIP_Retrieval, blip_retrieval  # noqa: F401
