import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from vidil_amd import kernels as k
from vidil_amd.kernels import kv_tile_offsets
DEV = "cuda"
def rand(*shape, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed); return torch.randn(*shape, generator=g)
def join(x3):
    D = x3.shape[-1] // 3; return x3[..., :D].float() + x3[..., D:2 * D].float()
for (B, nb, Nq, Nk) in [(5, 3, 1, 197), (4, 3, 4, 197), (3, 1, 4, 50), (2, 3, 1, 577), (4, 3, 2, 197), (4, 1, 12, 197), (4,3,4,64)]:
    H = 12; C = H * 64; Tc = (Nk + 31) // 32 * 32
    q = rand(B * nb * Nq, 3 * C, seed=80).to(DEV)[:, C:2 * C]
    kf, vf = rand(B, H, Nk, 64, seed=81), rand(B, H, Nk, 64, seed=82)
    k16, v16 = kf.half(), vf.half()
    ko, vo = kv_tile_offsets(Nk)
    kt = torch.zeros(B, H, Tc * 64, dtype=torch.float16); vt = torch.zeros(B, H, Tc * 64, dtype=torch.float16)
    kt[:, :, ko.reshape(-1)] = k16.reshape(B, H, -1); vt[:, :, vo.reshape(-1)] = v16.reshape(B, H, -1)
    out3 = torch.zeros(B * nb * Nq, 3 * C, dtype=torch.float16, device=DEV)
    k.attention_f32(q, kt.view(B, H, Tc, 64).to(DEV), vt.view(B, H, Tc, 64).to(DEV), out3, Bq=B * nb, H=H, Nq=Nq, Nk=Nk, kv_rows=Tc, kv_group=nb, arith=1, kv16=True)
    got = join(out3.cpu()).double()
    def ref(qq, pround=False):
        qd = qq.double().cpu().view(B * nb, Nq, H, 64).permute(0, 2, 1, 3)
        kd = k16.double().repeat_interleave(nb, 0); vd = v16.double().repeat_interleave(nb, 0)
        s = (qd @ kd.transpose(-1, -2)) * 0.125
        if pround:
            e = torch.exp(s - s.amax(-1, keepdim=True)); p_ = e.half().double() / e.sum(-1, keepdim=True)
        else:
            p_ = torch.softmax(s, -1)
        return (p_ @ vd).permute(0, 2, 1, 3).reshape(B * nb * Nq, C)
    e_exact = (got - ref(q)).abs()
    e_q16 = (got - ref((q * 0.125).half().float() * 8)).abs().max().item()
    e_p16 = (got - ref(q, True)).abs().max().item()
    rows = e_exact.view(B * nb * Nq, H, 64).amax(-1)      # per row, head
    print((B, nb, Nq, Nk), f"exact {e_exact.max().item():.2e} q16-ref {e_q16:.2e} p16-ref {e_p16:.2e}; per-row-in-batch max:", [f"{x:.1e}" for x in rows.view(B * nb, Nq, H).amax(-1).amax(0).tolist()])
    bad = (e_exact > 2e-6).nonzero()
    D = C
    for (r, c) in bad[:6].tolist():
        print("   bad at row", r, "col", c, "(head", c // 64, "d", c % 64, ") got", got[r, c].item(), "ref", ref(q)[r, c].item(), "hi", out3[r, c].item(), "lo", out3[r, D + c].item(), "hi2", out3[r, 2 * D + c].item())
    print("   n bad", bad.shape[0])
