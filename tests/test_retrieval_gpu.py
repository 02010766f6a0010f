"""GPU parity of the BLIP retrieval backend (--encoder_version blip): dense exact-f32 scores and row top-k kernels
bit-exact against the oracle, BLIP_Retrieval features / re-rank scores against the fp32 restatement, and the
per-frame top-k through BlipVisualTokenizer."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from common import ROOT, perturb_, synthetic_frames

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_scan_scores_bit_exact_and_topk_rows():
    from vidil_amd import kernels as K

    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libscan_ref.so"))
    lib.vidil_ref_score.restype = ctypes.c_float
    g = torch.Generator().manual_seed(3)
    NF, NC, D = 37, 421, 256
    img = torch.randn(NF, D, generator=g); img /= img.norm(dim=-1, keepdim=True)
    txt = torch.randn(NC, D, generator=g); txt /= txt.norm(dim=-1, keepdim=True)
    txt[100] = txt[7]; txt[300] = txt[7]                                   # exact ties
    out = K.scan_scores(img.to(DEV), txt.to(DEV)).cpu().numpy()
    im, tx = np.ascontiguousarray(img.numpy()), np.ascontiguousarray(txt.numpy())
    ref = np.empty((NF, NC), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    for f in range(NF):
        for c in range(NC):
            ref[f, c] = lib.vidil_ref_score(tx[c].ctypes.data_as(fp), im[f].ctypes.data_as(fp), D)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))       # bit-exact
    for k in (1, 5, 128):
        v, i = K.topk_rows(torch.from_numpy(out).to(DEV), k)
        order = np.lexsort((np.broadcast_to(np.arange(NC), ref.shape), -ref.astype(np.float64)), axis=1)[:, :k]
        assert np.array_equal(i.cpu().numpy(), order)
        assert np.array_equal(v.cpu().numpy(), np.take_along_axis(ref, order, 1))
    # strided rows (a column slice) and the error path
    sub = torch.from_numpy(out).to(DEV)[:, 16:216]
    v, i = K.topk_rows(sub, 4)
    assert np.array_equal(i.cpu().numpy(), np.lexsort((np.broadcast_to(np.arange(200), (NF, 200)), -ref[:, 16:216].astype(np.float64)), axis=1)[:, :4])
    with pytest.raises(K.VidilHipError):
        K.topk_rows(torch.zeros(2, 50000, device=DEV), 3)


def _small_retrieval(tmp_path):
    from vidil_amd.blip_retrieval import BLIP_Retrieval
    from vidil_amd.tokenizer import SyntheticBertTokenizer
    from vidil_amd.vit import VisionTransformer

    cfg = dict(architectures=["BertModel"], hidden_act="gelu", hidden_size=256, initializer_range=0.02, intermediate_size=512,
               layer_norm_eps=1e-12, max_position_embeddings=64, model_type="bert", num_attention_heads=4,
               num_hidden_layers=2, pad_token_id=0, type_vocab_size=2, vocab_size=30524, encoder_width=256,
               add_cross_attention=True)
    path = os.path.join(tmp_path, "med_small.json")
    json.dump(cfg, open(path, "w"))
    torch.manual_seed(11)
    m = BLIP_Retrieval(med_config=path, image_size=64, vit="base", embed_dim=64, tokenizer=SyntheticBertTokenizer())
    # shrink the vision tower to the small geometry (create_vit only knows base/large)
    m.visual_encoder = VisionTransformer(img_size=64, patch_size=16, embed_dim=256, depth=2, num_heads=4)
    m.vision_proj = torch.nn.Linear(256, 64)
    from vidil_amd.med import BertModel
    tcfg = m.text_encoder.config
    tcfg.encoder_width = 256
    m.text_encoder = BertModel(config=tcfg, add_pooling_layer=False)     # cross-attention K/V sized for the small ViT
    return m.eval()


def test_blip_retrieval_features_rerank_and_tokenizer_vs_oracle(tmp_path):
    from oracle import clip_ref, retrieval_ref
    from vidil_amd.visual_tokenization import CATEGORIES, BlipVisualTokenizer

    m = _small_retrieval(str(tmp_path))
    perturb_(m, 700)
    with torch.no_grad():                                       # spread the similarities and the ITM logits
        m.vision_proj.weight.mul_(8); m.text_proj.weight.mul_(8); m.itm_head.weight.mul_(20)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    NF, k_test, topk = 5, 6, 3
    u8 = synthetic_frames(1, NF, size=64, first_video=61)[0]
    x = clip_ref.preprocess_u8(u8)
    texts = {key: [f"w{1000 + 37 * c + j} w{2000 + j}" if j % 3 else f"w{3000 + 11 * c + j}" for j in range(n)]
             for c, (key, n) in enumerate(zip(CATEGORIES, (23, 17, 9, 12)))}
    # the reference embeds / re-ranks the PROMPTED strings ('v1': "A photo of {x}", run_visual_tokenization.py:199-201)
    # for encoder_version blip too, and emits the raw class strings
    tok = BlipVisualTokenizer(dict(topk_visualize=topk, k_test=k_test, image_size=64,
                                   prompt_version_visual_tokenization="v1"), m, texts, DEV)
    with torch.no_grad():
        y_ref, img_ref = retrieval_ref.image_features(sd, x, depth=2, heads=4)
    y16, img = m.image_features_u8(torch.from_numpy(u8).to(DEV))
    assert (img.cpu() - img_ref).abs().max().item() < 2e-3
    idx, score = tok.frame_topk(torch.from_numpy(u8).to(DEV))
    idx, score = idx.cpu().numpy(), score.cpu().numpy()
    for c, key in enumerate(CATEGORIES):
        ids, lens = m.tokenize([f"A photo of {t}" for t in texts[key]])
        mask = (torch.arange(35)[None] < lens[:, None]).long()
        with torch.no_grad():
            txt_ref = retrieval_ref.text_features(sd, ids.long(), mask, layers=2, H=4)
        assert (tok.text_repr[key]["embeds"].cpu() - txt_ref).abs().max().item() < 2e-3
        ids_enc = ids.clone().long(); ids_enc[:, 0] = m.tokenizer.enc_token_id
        assert torch.equal(tok.text_repr[key]["ids"].cpu().long(), ids_enc)
        with torch.no_grad():
            sims, full = retrieval_ref.score_matrix(sd, y_ref, img_ref, txt_ref, ids_enc, mask, k_test, layers=2, H=4)
        full = full.numpy()
        for f in range(NF):
            order = np.argsort(-full[f], kind="stable")[:topk]
            ref_s = full[f][order]
            assert np.abs(score[f, c] - ref_s).max() < 2e-2, (key, f, score[f, c], ref_s)
            gaps = np.abs(np.diff(np.sort(full[f])[::-1][:topk + 1]))
            # candidate membership (rank k_test vs k_test+1 of the similarities) and order are only compared where the
            # oracle's own margins exceed the f16 error of the towers
            sim_sorted = np.sort(sims[f].numpy())[::-1]
            if gaps.min() > 4e-2 and sim_sorted[k_test - 1] - sim_sorted[k_test] > 5e-3:
                assert list(idx[f, c]) == list(order), (key, f, idx[f, c], order)
    out = tok.process(["v0"], torch.from_numpy(u8[None]).to(DEV), [["cap"]])
    assert set(out["v0"]["frame_tokens"][0].keys()) == set(CATEGORIES) and len(out["v0"]["frame_tokens"]) == NF
    for key in CATEGORIES:                                       # emitted tokens are the unprompted class strings
        assert all(t in texts[key] for fr in out["v0"]["frame_tokens"] for t in fr[key])


def test_blip_itm_forward_with_the_itc_head_vs_oracle(tmp_path):
    """BLIP_ITM.forward(image, caption, match_head='itc') (models/blip_itm.py:60-67; round 5: it used to raise on BLIP_ITM and exist
    on BLIP_Retrieval only): normalize(vision_proj(image [CLS])) @ normalize(text_proj(text [CLS], mode='text'))^T."""
    from oracle import clip_ref, retrieval_ref
    from vidil_amd.blip_itm import BLIP_ITM

    m = _small_retrieval(str(tmp_path))
    perturb_(m, 701)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    NF = 4
    u8 = synthetic_frames(1, NF, size=64, first_video=62)[0]
    x = clip_ref.preprocess_u8(u8)
    caps = [f"w{1200 + 7 * j} w{2100 + j} w{3300 + 3 * j}" for j in range(NF)]
    sim = BLIP_ITM.forward(m, x.to(DEV), caps, match_head="itc")          # (the base class's own 'itc' branch)
    assert tuple(sim.shape) == (NF, NF)
    assert torch.equal(sim, m.forward(x.to(DEV), caps, match_head="itc"))
    ids, lens = m.tokenize(caps)
    mask = (torch.arange(35)[None] < lens[:, None]).long()
    with torch.no_grad():
        _, img_ref = retrieval_ref.image_features(sd, x, depth=2, heads=4)
        txt_ref = retrieval_ref.text_features(sd, ids.long(), mask, layers=2, H=4)
    assert (sim.cpu() - img_ref @ txt_ref.t()).abs().max().item() < 3e-3
    with pytest.raises(ValueError):
        BLIP_ITM.forward(m, x.to(DEV), caps, match_head="nope")
