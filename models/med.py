"""Shim for ``from models.med import ...`` -> vidil_amd.med."""
from vidil_amd.med import BertConfig, BertLMHeadModel, BertModel  # noqa: F401
