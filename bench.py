#!/usr/bin/env python
"""bench.py — frames/sec of VidIL's frame-encoding hot path on MI355X.

One "step" = one batch of synthetic videos (default 384 videos x 8 frames, 224^2 uint8,
already resident in HBM) through the WHOLE path: BLIP ViT-B/16 caption (beam 3,
max_length 20) + CapFilt ITM filter + CLIP ViT-B/32 visual tokens against a vg-sized
ontology (42,759 classes), including the host-side string work and the device->host
copies of the results.  Weights are random-init (seed 0) of the named architectures:
captions never reach [SEP], so every frame pays the worst case of 16 decode steps and
8 unique captions per video go through the filter.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see README / DESIGN.md §measurement).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VG_SIZES = dict(objects=19958, attributes=15026, scenes=365, verbs=7410)
GFLOP_PER_FRAME = dict(vit_caption=35.13, vit_filter=35.13, decode=20.05, itm_kv=5.58, itm_per_caption=7.24,
                       clip=8.82, scan=0.044)
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md


def synthetic_frames(n_videos, frames, size, first_video=0):
    out = np.empty((n_videos, frames, size, size, 3), dtype=np.uint8)
    for v in range(n_videos):
        rng = np.random.default_rng(1000 + first_video + v)
        out[v] = rng.integers(0, 256, size=(frames, size, size, 3), dtype=np.uint8)
    return out


def synthetic_ontology(dim=512, seed=0):
    g = torch.Generator().manual_seed(seed)
    embeds, texts = {}, {}
    for k, n in VG_SIZES.items():
        e = torch.randn(n, dim, generator=g)
        e = e / e.norm(dim=-1, keepdim=True)
        t = [f"{k}_{i}" for i in range(n)]
        if k == "scenes":   # the real list carries duplicate strings ('outdoor' x25, 'indoor' x25 ...)
            for j in range(1, 25):
                e[40 + j] = e[40]; t[40 + j] = t[40]
                e[100 + j] = e[100]; t[100 + j] = t[100]
        embeds[k], texts[k] = e, t
    return embeds, texts


def build_models(device, size=224, clip_name="b32"):
    from vidil_amd.blip import BLIP_Decoder
    from vidil_amd.blip_itm import BLIP_ITM
    from vidil_amd.clip import CLIPConfig, CLIPModel
    from vidil_amd.tokenizer import SyntheticBertTokenizer

    torch.manual_seed(0)
    tok = SyntheticBertTokenizer()
    cap = BLIP_Decoder(image_size=size, vit="base", tokenizer=tok).eval()
    flt = BLIP_ITM(image_size=size, vit="base", tokenizer=tok).eval()
    clip = CLIPModel(CLIPConfig.vit_l14() if clip_name == "l14" else None).eval()
    return cap, flt, clip, tok


class GemmTimer:
    """Times every GEMM launch of one step with HIP events on the launch stream."""

    def __init__(self):
        self.records = []

    @staticmethod
    def tile_of(M, N, K=768, f16_out=True):
        """Mirror of the dispatch in csrc/gemm.hip::pick_tile / gemm256.hip::vidil_gemm256_eligible."""
        t256 = ((M + 255) // 256) * ((N + 255) // 256)
        if t256 >= 160 and K >= 128 and N % (8 if f16_out else 4) == 0:
            return "256x256"
        n128 = ((M + 127) // 128) * ((N + 127) // 128)
        if n128 >= 200 and N <= 1024:
            return "128x128x2" if n128 > 256 else "128x128x3"
        if ((M + 63) // 64) * ((N + 63) // 64) <= 1280:
            return "64x64x3"
        if ((M + 127) // 128) * ((N + 63) // 64) <= 2560:
            return "128x64x2"
        return "128x128x2"

    def install(self):
        from vidil_amd import kernels as K

        self._orig = K.gemm
        timer = self

        def timed(a, w, bias=None, **kw):
            M = kw.get("M") if kw.get("lda") is not None else a.shape[0]
            N, Kd = w.shape
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = timer._orig(a, w, bias, **kw)
            e1.record()
            epi = "heads" if kw.get("heads") else "patch" if kw.get("patch") else "arena" if kw.get("arena") else \
                ("f32" if (kw.get("out") is not None and kw["out"].dtype == torch.float32) or kw.get("out_dtype") == torch.float32 else "f16")
            tile = timer.tile_of(M, N, Kd, epi in ("f16", "heads"))
            if epi == "arena" and tile == "256x256":      # EPI_ARENA is served by the small-tile kernel only
                tile = timer.tile_of(M, N, 0)
            timer.records.append((tile, epi, kw.get("act", 0),
                                  2.0 * M * N * Kd, e0, e1))
            return r

        K.gemm = timed

    def remove(self):
        from vidil_amd import kernels as K

        K.gemm = self._orig

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        epi_id = dict(f16=0, f32=1, heads=2, patch=3, arena=4)
        for tile, epi, act, flops, e0, e1 in self.records:
            # same spelling as the kernel names in the rocprofv3 trace (profiles/*.md)
            if tile == "256x256":
                key = f"gemm256_kernel<{epi_id[epi]}, {act}>"
            else:
                bm, bn, st = tile.split("x")
                key = f"gemm_kernel<{bm}, {bn}, {st}, {epi_id[epi]}, {act}>"
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += e0.elapsed_time(e1) * 1e-3
        return agg


def cpu_baseline(cap, flt, clip, tok, onto_embeds, onto_texts, n_videos, frames, size):
    """The CPU oracle in the reference's schedule on a bounded sample (rank 0, N=1 only)."""
    from oracle import clip_ref, pipeline_ref

    ncpu = min(len(os.sched_getaffinity(0)), int(os.environ.get("VIDIL_CPU_THREADS", "64")))
    torch.set_num_threads(ncpu)
    sd_cap = {k: v.detach().float().cpu() for k, v in cap.state_dict().items()}
    sd_itm = {k: v.detach().float().cpu() for k, v in flt.state_dict().items()}
    sd_clip = {k: v.detach().float().cpu() for k, v in clip.state_dict().items()}
    prompt = cap.prompt_ids(1, "cpu")[0].long().numpy()
    fr = synthetic_frames(n_videos, frames, size)
    t0 = time.time()
    n_caps = 0
    for v in range(n_videos):
        x = clip_ref.preprocess_u8(fr[v])
        kept, caps = pipeline_ref.capfilt_video(sd_cap, sd_itm, x, prompt, tok, cap.prompt, threshold=0.4)
        n_caps += len(caps)
        pipeline_ref.visual_tokens_video(sd_clip, x, onto_embeds, onto_texts, topk=5)
    dt = time.time() - t0
    return dict(value=round(n_videos * frames / dt, 4), unit="frames/s", cores=ncpu, kind="port",
                sample=f"{n_videos} video(s) x {frames} frames, oracle (PyTorch fp32, {torch.get_num_threads()} threads) in the "
                       f"reference's schedule incl. {n_caps} ITM caption passes, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    # 384 videos = 3072 frames per step: 605,184 ViT rows = 2364 row tiles of 256, i.e. 27.7 rounds of 256-tile
    # workgroups on the N = 768 GEMMs (98.9 % of the last round filled; 128 videos give 9.23 rounds = 92 %), and
    # 9216 beam rows per decode step (measured: 128 -> 3.7k, 192 -> 3.8-3.95k, 384 -> 4.1k, 576 -> 3.9k frames/s)
    ap.add_argument("--videos-per-step", type=int, default=384)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=224, help="frame / BLIP image size (the headline metric is 224)")
    ap.add_argument("--clip", choices=["b32", "l14"], default="b32", help="CLIP tower (headline metric: ViT-B/32)")
    ap.add_argument("--cpu-sample-videos", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    from vidil_amd import dist as vdist
    from vidil_amd.capfilt import CapFiltEngine
    from vidil_amd.visual_tokenization import VisualTokenizer

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback for the product path)")
    # (developer smoke of the N > 1 launch path on a one-GPU box: VIDIL_BENCH_SMOKE_ONE_DEVICE=1 puts every rank
    #  on cuda:0 and rendezvous over gloo; the real multi-GPU run is one rank per GPU over RCCL)
    one_device = os.environ.get("VIDIL_BENCH_SMOKE_ONE_DEVICE") == "1"
    backend = "gloo" if one_device else "nccl"
    rank, world, local = vdist.init_distributed_mode(backend=backend) if args.gpus > 1 else (0, 1, 0)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run")
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    t_start = time.perf_counter()
    cap, flt, clip, tok = build_models(dev, args.size, args.clip)
    onto_embeds, onto_texts = synthetic_ontology(dim=clip.config.projection_dim)
    headline = args.size == 224 and args.clip == "b32"
    config = dict(caption=True, filter=True, filter_generated_only=True, keep_original_caption=False,
                  threshold=0.4, filter_mode="max_filter", generation_mode="beam", do_sentence_tokenization=True,
                  image_size=args.size, vit="base", topk_visualize=5)
    engine = CapFiltEngine(config, dev, captioner=cap, filterer=flt)
    vtok = VisualTokenizer(config, clip, onto_texts, onto_embeds, dev)

    Nv, F = args.videos_per_step, args.frames
    first = rank * Nv
    frames = torch.from_numpy(synthetic_frames(Nv, F, args.size, first)).to(dev)
    video_ids = [f"video{first + i}" for i in range(Nv)]

    def step():
        items = [dict(video_id=v, text=[]) for v in video_ids]
        engine.process(items, frames)
        toks = vtok.process(video_ids, frames, [it["unfiltered_text"] for it in items])
        return items, toks

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    log("models + inputs ready")
    # one-time initialisation, like weight packing: the first batch of a shape runs the decode steps eagerly, the
    # second captures them into HIP graphs; steady state starts with the third (independent of --warmup)
    # ... and a fresh box needs a few seconds of load before clocks and caches settle: keep priming (at most 8
    # steps) until two consecutive steps agree within 2 %.
    prev = None
    for i in range(8):
        torch.cuda.synchronize()
        t_p = time.perf_counter()
        step()
        torch.cuda.synchronize()
        t_p = time.perf_counter() - t_p
        if i >= 2 and prev is not None and abs(t_p - prev) <= 0.02 * prev:
            break
        prev = t_p
    log(f"sessions primed after {i + 1} steps (decode-step graphs captured, last step {t_p * 1e3:.1f} ms)")
    for _ in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log("warmup step done")
    torch.cuda.synchronize()
    vdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        items, toks = step()
    torch.cuda.synchronize()
    vdist.barrier()
    torch.cuda.synchronize()
    dt = vdist.max_over_ranks(time.perf_counter() - t0)
    stats = dict(engine.last_stats)
    log(f"timed region done: {dt:.3f}s for {args.steps} steps")

    result = None
    if rank == 0:
        total_frames = world * Nv * F * args.steps
        fps = total_frames / dt
        c_mean = stats["unique_captions"] / max(1, stats["videos"])
        gf = GFLOP_PER_FRAME
        gflop_frame = (gf["vit_caption"] + gf["vit_filter"] + gf["decode"] + gf["itm_kv"] + gf["itm_per_caption"] * c_mean
                       + gf["clip"] + gf["scan"])
        result = {
            "metric": f"frames/sec whole-node (BLIP caption+filt + CLIP visual-token) {args.size}^2 8f/video",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{Nv} synthetic videos x {F} frames {args.size}^2 per GPU per step, BLIP ViT-B/16 caption "
                                   f"(beam 3, 16 decode steps) + CapFilt ITM + CLIP {'ViT-B/32' if args.clip == 'b32' else 'ViT-L/14'} visual tokens vs 42,759-class "
                                   f"vg-sized ontology; random-init weights (seed 0)",
                       "videos_per_step_per_gpu": Nv, "frames_per_video": F,
                       "unique_captions_per_video": round(c_mean, 2), "itm_pairs_per_step": stats["itm_pairs"],
                       "algorithmic_gflop_per_frame": round(gflop_frame, 2) if headline else None,
                       "whole_path_mfma_frac": round(fps / world * gflop_frame / 1e3 / MFMA_F16_PEAK_TFLOPS, 4) if headline else None,
                       "parallelism": f"dp{world} (videos sharded, no data-path collective)"},
        }
    if rank == 0 and world == 1 and not args.no_roofline:
        timer = GemmTimer()
        timer.install()
        # (the instrumented step launches the decode steps one by one instead of replaying their graphs, so that
        #  every GEMM launch passes through the timer, as in the rocprofv3 trace)
        states = list(getattr(cap, "_decode_state", {}).values())
        for st in states:
            st["graphs_ok"] = False
        step()
        for st in states:
            st["graphs_ok"] = True
        timer.remove()
        agg = timer.summary()
        key = max(agg, key=lambda k: agg[k][2])
        n, flops, secs = agg[key]
        ach = flops / secs / 1e12
        # HBM bytes per launch of that kernel from the committed PMC passes of this same command
        # (tools/profile_bench.sh -> profiles/pmc_traffic.json); null when no profile has been taken.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = pmc["kernels"].get(key, {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError, KeyError):
            pass
        result["roofline"] = {"bound": "mfma", "kernel": key, "achieved": round(ach, 1), "peak": MFMA_F16_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4), "traffic": traffic,
                              "algorithmic_flop_per_launch": round(flops / n),
                              "launches_per_step": n, "avg_launch_us": round(secs / n * 1e6, 2),
                              "all_gemm": {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1),
                                               "ms": round(v[2] * 1e3, 3)} for k, v in agg.items()}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline...")
        result["cpu_baseline"] = cpu_baseline(cap, flt, clip, tok, onto_embeds, onto_texts, args.cpu_sample_videos, F, args.size)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
