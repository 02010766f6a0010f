"""A/B of the persistent staged attention kernel against the one-workgroup-per-unit form: bit-equality and time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vidil_amd import kernels as K
dev = "cuda"


def run(B, H, T, dt, NP0=True, seed=0):
    torch.manual_seed(seed)
    q = (torch.randn(B, H, T, 64, device=dev) * 0.125).to(dt)
    k = torch.randn(B, H, T, 64, device=dev).to(dt)
    if NP0:
        v = torch.randn(B, H, T, 64, device=dev).to(dt); NP = 0
    else:
        NP = (T + 15) // 16 * 16
        v = torch.randn(B, H, 64, NP, device=dev).to(dt)
    outs = []
    for j in ("0", "1"):
        os.environ["VIDIL_ATTN_PERSIST"] = j
        o = torch.zeros(B * T, H * 64, dtype=dt, device=dev)
        K.attention(q, k, v, o, Bq=B, H=H, Nq=T, Nk=T, Tq_cap=T, Tk_cap=T, NP=NP)
        torch.cuda.synchronize()
        outs.append(o.clone())
    eq = torch.equal(outs[0], outs[1])
    err = float("nan")
    if NP0:
        att = torch.softmax(q.float() @ k.float().transpose(-1, -2), dim=-1) @ v.float()
        err = (outs[1].float() - att.permute(0, 2, 1, 3).reshape(B * T, H * 64)).abs().max().item()
    print(f"B={B} H={H} T={T} {str(dt)[6:]} NP0={NP0}: persistent == per-unit form: {eq}; max|persistent - torch| = {err:.2e}")
    assert eq


for dt in (torch.float16, torch.bfloat16):
    run(100, 12, 197, dt)
    run(300, 4, 129, dt)
    run(70, 16, 256, dt)
    run(97, 12, 161, dt, NP0=False)
    run(2000, 12, 197, dt)
print("ok")
