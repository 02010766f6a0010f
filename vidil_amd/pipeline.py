"""One batch of videos through BOTH frame-encoding scripts with the GPU queue kept full.

The reference runs ``run_video_CapFilt.py`` and then ``run_visual_tokenization.py`` over the dataset, each video by
itself, each GPU call followed by host string work (tokenizer decode, de-duplication, ontology lookups) during which
the device idles.  Per batch the data dependencies are only

    frames -> caption ViT -> beam decode -> [host: ids -> strings -> distinct captions -> ids] -> ITM (needs filter ViT)
    frames -> filter ViT                                                                        /
    frames -> CLIP tower -> ontology scan / top-k -> [host: indices -> strings -> aggregation]

so this driver queues the three towers back to back and does each piece of host work behind an event while a later
tower is still running (a kernel trace of the sequential order shows ~35 ms of idle device per 3,072-frame batch,
5 % of the step).  One stream, no extra memory beyond the filter ViT's output staying alive (0.9 GB of 288).
The results are those of ``CapFiltEngine.process`` + ``VisualTokenizer.process`` called one after the other
(tests/test_models_gpu.py checks equality).
"""
from __future__ import annotations

import torch


class FramePipeline:
    def __init__(self, engine, visual_tokenizer):
        self.engine = engine
        self.vtok = visual_tokenizer

    @torch.no_grad()
    def process(self, items, frames_u8):
        """items: [{'video_id', 'text'}]; frames_u8 uint8 [Nv,F,H,W,3] on the device.
        Returns (items with 'text' / 'unfiltered_text' filled, {video_id: visual-token dict})."""
        eng, vt = self.engine, self.vtok
        F = frames_u8.shape[1]
        st = eng.begin(items, frames_u8)             # caption ViT + decode; ids on their way to the host
        pending_idx = vt.begin(frames_u8)            # CLIP tower + scan + top-k; indices on their way to the host
        eng.encode_filter_frames(st)                 # filter ViT
        eng.captions_ready(st)                       # host: strings, distinct captions; queues the ITM pairs
        video_ids = [it["video_id"] for it in items]
        tokens = vt.assemble(video_ids, pending_idx, [it["unfiltered_text"] for it in items], F)   # under the ITM
        eng.finish(st)
        return items, tokens
