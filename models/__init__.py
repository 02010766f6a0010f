"""Import-path shim: the reference's scripts do ``from models.blip import blip_decoder`` /
``from models.blip_itm import blip_itm`` (run_video_CapFilt.py:11-12).  With this repository on
PYTHONPATH those imports resolve to the MI355X implementation in ``vidil_amd``."""
