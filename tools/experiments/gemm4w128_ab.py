"""Run a fixed set of GEMMs and print a digest and the time of each: executed once with VIDIL_GEMM4W128=0 and once with
=1 (the variable is read once per process), the two outputs must show the same digests (bit-identical kernels)."""
import hashlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vidil_amd import kernels as K
dev = "cuda"


def timeit(fn):
    t0 = time.time()
    while time.time() - t0 < 0.1:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3


def digest(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()[:12]


def rand(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


H = 12
for dt in (torch.float16, torch.bfloat16):
    for (M, N, Kd) in [(10752, 768, 768), (10752, 768, 3072), (10752, 3072, 768), (9999, 832, 512), (30000, 768, 128), (128 * 70 + 5, 256 * 3, 192)]:
        a = rand(M, Kd, seed=1).to(dt).to(dev)
        w = rand(N, Kd, scale=0.05, seed=2).to(dt).to(dev)
        bias = rand(N, seed=3).to(dev)
        x0 = rand(M, N, seed=4).to(dev)
        name = K.gemm_kernel_name(a, w, bias, out=x0, resid=x0).split("<")[0] + "/" + K.gemm_kernel_name(a, w, bias, out=x0, resid=x0).split(",")[-1]
        x = x0.clone(); K.gemm(a, w, bias, out=x, resid=x)
        o1 = K.gemm(a, w, bias, act=K.ACT_GELU_ERF)
        o2 = K.gemm(a, w, None)
        o3 = K.gemm(a, w, bias, act=K.ACT_QUICK_GELU)
        xs = x0.clone()
        t_f32 = timeit(lambda: K.gemm(a, w, bias, out=xs, resid=xs))
        t_gelu = timeit(lambda: K.gemm(a, w, bias, act=K.ACT_GELU_ERF))
        extra = ""
        if N == H * 64:
            q = torch.zeros(M, H, 1, 64, dtype=dt, device=dev)
            hd = dict(q=q, T=1, H=H, part0=0, t_off=0, Tq_cap=1, q_scale=0.125)
            K.gemm(a, w, bias, heads=hd)
            extra = " heads " + digest(q)
        print(f"{str(dt)[6:]:8s} M={M:6d} N={N:5d} K={Kd:5d} {name:28s} f32+res {digest(x)} {t_f32:7.1f} us | gelu {digest(o1)} {t_gelu:7.1f} us | plain {digest(o2)} quick {digest(o3)}{extra}")
