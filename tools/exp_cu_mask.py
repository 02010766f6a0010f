"""developer experiment: the caption decode (latency / HBM bound, far below the power cap) side by side with a ViT
(MFMA bound, at the power cap) on DISJOINT CU sets of one MI355X (hipExtStreamCreateWithCUMask).

    python tools/exp_cu_mask.py            (GPU box)

Prints: ViT time on n CUs alone, decode time on n CUs alone, and both together for a few splits.
"""
import ctypes
import os
os.environ.setdefault("VIDIL_DEV_ENV", "1")   # the library caches its developer switches per process otherwise
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models, synthetic_frames  # noqa: E402
from vidil_amd.blip import CLIP_MEAN, CLIP_STD  # noqa: E402

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for i in bits:
        words[i >> 5] |= 1 << (i & 31)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    dev = torch.device("cuda", 0)
    dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    cap, flt, clip, tok = build_models(dev, 224, "b32", "base", dt)
    cap, flt = cap.to(dev), flt.to(dev)
    Nv, F = 384, 8
    frames = torch.from_numpy(synthetic_frames(Nv, F, 224, 0)).to(dev).reshape(Nv * F, 224, 224, 3)

    def vit():
        return flt.visual_encoder.forward_u8(frames, CLIP_MEAN, CLIP_STD)[1]

    y16 = cap.visual_encoder.forward_u8(frames, CLIP_MEAN, CLIP_STD)[1]

    def decode():
        return cap.generate_ids(y16, Nv * F, num_beams=3, max_length=20, min_length=5)[0]

    for _ in range(3):          # eager, capture, replay
        decode(); vit()
    torch.cuda.synchronize()

    def timed(fn, stream, cus, reps=3):
        os.environ["VIDIL_GEMM_CUS"] = str(cus)
        with torch.cuda.stream(stream):
            fn()
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            stream.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    full = torch.cuda.current_stream()
    print(f"dtype {dt}; unmasked: ViT {timed(vit, full, 256):.1f} ms, decode {timed(decode, full, 256):.1f} ms", flush=True)
    for n in (256, 224, 192, 160, 128):
        s = masked_stream(range(n))
        print(f"ViT on the first {n} mask bits: {timed(vit, s, n):.1f} ms", flush=True)
    for n in (32, 64, 96, 128):
        s = masked_stream(range(256 - n, 256))
        print(f"decode on the last {n} mask bits: {timed(decode, s, n):.1f} ms", flush=True)
    for n_dec in (0, 32, 64, 96):
        if n_dec:
            sv, sd = masked_stream(range(256 - n_dec)), masked_stream(range(256 - n_dec, 256))
        else:
            sv, sd = torch.cuda.Stream(), torch.cuda.Stream()       # both unmasked: whatever the hardware schedules
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        ev = []
        for _ in range(reps):
            os.environ["VIDIL_GEMM_CUS"] = str(256 - n_dec)
            with torch.cuda.stream(sv):
                vit()
            os.environ["VIDIL_GEMM_CUS"] = str(max(8, n_dec) if n_dec else 256)
            with torch.cuda.stream(sd):
                decode()
        with torch.cuda.stream(sv):
            e1 = torch.cuda.Event(enable_timing=False); e1.record()
        sv.synchronize()
        t_v = time.perf_counter() - t0
        sd.synchronize()
        t_all = time.perf_counter() - t0
        print(f"together, decode on {n_dec or 'no mask'} CUs: ViT stream done after {t_v / reps * 1e3:.1f} ms/rep, both after "
              f"{t_all / reps * 1e3:.1f} ms/rep (serial sum unmasked above)", flush=True)


if __name__ == "__main__":
    main()
