"""vidil_amd — MI355X-native frame-encoding hot path of VidIL.

Only what the path needs lives here: ``csrc/`` (HIP kernels + C ABI) and the
host-side mirror of the reference's Python interface (``blip_decoder``,
``blip_itm`` / ``BLIP_ITM``, the CLIP model object, the two driver functions).
"""
__version__ = "0.1.0"
