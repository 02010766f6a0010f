import os, sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from common import perturb_, synthetic_frames
from oracle import beam_ref, clip_ref, med_ref, vit_ref
from vidil_amd.blip import BLIP_Decoder
from vidil_amd.packing import set_compute_dtype, set_parity_mode
from vidil_amd.tokenizer import SyntheticBertTokenizer
torch.manual_seed(0)
cap = BLIP_Decoder(image_size=224, vit="base", tokenizer=SyntheticBertTokenizer()).eval(); perturb_(cap, 100)
sd = {k: v.clone() for k, v in cap.state_dict().items()}
cap = cap.to("cuda"); set_compute_dtype("f16", cap); set_parity_mode(True, cap)
B, nb = 6, 3
u8 = synthetic_frames(1, B)[0]
with torch.no_grad(): y_ref = vit_ref.vit_forward(sd, clip_ref.preprocess_u8(u8))
enc3 = y_ref.repeat_interleave(nb, 0); state = {}; otrace = []
def step(ids, beam_idx):
    with torch.no_grad():
        past = None if beam_idx is None else med_ref.reorder_cache(state["cache"], torch.from_numpy(beam_idx))
        lg, state["cache"] = med_ref.decoder_logits(sd, torch.from_numpy(ids), enc3, past)
    return lg.numpy()
prompt = cap.prompt_ids(B, "cpu").long().numpy()
seqs, _ = beam_ref.beam_search(step, prompt, num_beams=nb, max_length=20, min_length=5, eos_token_id=102, pad_token_id=0, trace=otrace)
gaps = np.stack([np.min(t["cand_scores"][:, :-1] - t["cand_scores"][:, 1:], axis=1) for t in otrace]).min(axis=0)
_, y3 = cap.visual_encoder.forward_u8(torch.from_numpy(u8).cuda(), clip_ref.CLIP_MEAN, clip_ref.CLIP_STD)
for rep in range(3):
    tok, _ = cap.generate_ids(y3, B, num_beams=nb, max_length=20, min_length=5)
    t = tok.cpu().numpy()
    print(os.environ.get("VIDIL_PARITY_ATTN","f32"), "rep", rep, [bool(np.array_equal(t[b][:len(seqs[b])], seqs[b])) for b in range(B)], ["%.1e" % g for g in gaps])
for b in range(B):
    if not np.array_equal(t[b][:len(seqs[b])], seqs[b]):
        first = int(np.argmax(t[b][:len(seqs[b])] != seqs[b]))
        print("image", b, "first differing position", first, "device", t[b][:len(seqs[b])].tolist(), "oracle", seqs[b].tolist())
        st = first - 4
        if 0 <= st < len(otrace): print("  oracle candidates at that step:", otrace[st]["cand_scores"][b], otrace[st]["cand_index"][b])
