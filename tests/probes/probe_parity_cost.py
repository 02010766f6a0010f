"""Developer probe: one pipeline step in a parity configuration, to be run under rocprofv3 --kernel-trace --stats.
usage: python tests/probes/probe_parity_cost.py <full|mix|plain> [videos]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vidil_amd.capfilt import CapFiltEngine  # noqa: E402
from vidil_amd.packing import set_parity_mode  # noqa: E402
from vidil_amd.pipeline import FramePipeline  # noqa: E402
from vidil_amd.visual_tokenization import VisualTokenizer  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "full"
Nv = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
cap, flt, clip, tok = bench.build_models(dev, dtype="f16")
emb, texts = bench.synthetic_ontology(dim=512)
cfg = dict(caption=True, filter=True, filter_generated_only=True, keep_original_caption=False, threshold=0.4, filter_mode="max_filter",
           generation_mode="beam", do_sentence_tokenization=False, image_size=224, vit="base", topk_visualize=5)
eng = CapFiltEngine(cfg, dev, captioner=cap, filterer=flt)
vt = VisualTokenizer(cfg, clip, texts, emb, dev)
frames = torch.from_numpy(bench.synthetic_frames(Nv, 8, 224, 0)).to(dev)
if mode == "full":
    set_parity_mode(True, cap, flt, clip)
elif mode == "mix":
    set_parity_mode(True, cap, clip)
    cap.visual_encoder.set_parity_last_blocks(0)
pipe = FramePipeline(eng, vt)


def step():
    return pipe.process([dict(video_id=f"video{i}", text=[]) for i in range(Nv)], frames)


for _ in range(2):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
step()
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"{mode}: {Nv * 8 / dt:.0f} frames/s ({dt * 1e3:.0f} ms per {Nv}-video step)")
