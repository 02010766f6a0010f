"""oracle/synth_weights.py — TEST INFRASTRUCTURE.  Synthetic "trained-like" weight statistics for the parity checks.

No BLIP checkpoint can be downloaded here (reference: models/blip.py:332-354 load_checkpoint,
docker/download_blip_checkpoints.sh:3-7), and random-init weights lack what makes an ABSOLUTE caption-logit tolerance hard at a
trained model's scale: max|logit| of 10-25 (random init: ~2.5), LayerNorm gains with outlier channels, residual rows off zero.
``trained_like_`` gives a random-init model those statistics, seeded; the fp32 oracle and the device run the same state dict.
Used by tests/ (tests/common.py re-exports it) and by bench.py's parity leg (the `parity_qualified` error at a trained scale)."""
import numpy as np
import torch


def trained_like_(module, seed, *, head_scale=7.0, outlier_gain=(10.0, 50.0), n_outliers=3, stream_shift=1.0, sep_bias=None):
    """Give a random-init model the statistics that trained ViT / BERT checkpoints have and random init lacks (VERDICT r3 #4;
    no weights can be downloaded here): in place, seeded.
      * every LayerNorm gain gets ``n_outliers`` channels multiplied by 10-50x (the outlier channels of trained transformers)
        and the 1-D parameters the N(0, 0.05) jitter of ``perturb_``;
      * the residual streams get rows whose mean is ~``stream_shift`` standard deviations away from zero: a constant added to
        the position embeddings (ViT: pos_embed; BERT: position_embeddings) — what the LayerNorm fold's error model calls mu / sigma;
      * the LM head (``cls.predictions.decoder.weight``) is scaled by ``head_scale``: random init gives max|logit| ~ 2.5,
        trained captioners 10-25 — an ABSOLUTE logit tolerance is a different claim at that scale;
      * ``sep_bias`` (optional) is added to the [SEP] logit bias so that beam searches end at different lengths.
    Returns a dict describing what was done (for the test's printout)."""
    g = torch.Generator().manual_seed(seed)
    done = dict(layernorms=0, outliers=0)
    with torch.no_grad():
        for p in module.parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
        for m in module.modules():
            if isinstance(m, torch.nn.LayerNorm):
                idx = torch.randperm(m.weight.numel(), generator=g)[:n_outliers]
                gain = outlier_gain[0] + (outlier_gain[1] - outlier_gain[0]) * torch.rand(n_outliers, generator=g)
                m.weight[idx] *= gain
                done["layernorms"] += 1
                done["outliers"] += n_outliers
        sd = dict(module.named_parameters())
        for name, p in sd.items():
            if name.endswith("pos_embed") or name.endswith("position_embeddings.weight"):
                p.add_(stream_shift * float(p.std()) * 8.0)      # (rows of the stream then sit ~stream_shift sigma off zero; measured in the test)
            if name.endswith("cls.predictions.decoder.weight"):
                p.mul_(head_scale)
            if sep_bias is not None and (name.endswith("cls.predictions.bias") or name.endswith("cls.predictions.decoder.bias")):
                p[102] += sep_bias
    return done


def portable_init_(module, seed):
    """Overwrite every floating-point parameter of ``module`` from numpy's PCG64 stream — bit-identical on every host.
    (torch's CPU normal sampler is vectorised differently on AVX2 and AVX-512 machines: `torch.manual_seed(0)` gives the build
    container and the GPU box DIFFERENT random-init weights, so goldens tied to seeded torch init do not travel.)
    Matrices / embeddings ~ N(0, 0.02) (the reference's init scale: models/vit.py:163-174, models/med.py:558-568), LayerNorm
    gains 1 + N(0, 0.05), every other vector N(0, 0.05); parameters are visited in ``named_parameters`` order."""
    rng = np.random.default_rng(seed)
    ln_gains = {id(m.weight) for m in module.modules() if isinstance(m, torch.nn.LayerNorm)}
    with torch.no_grad():
        for name, p in module.named_parameters():
            if not p.is_floating_point():
                continue
            if p.ndim >= 2:
                v = rng.standard_normal(p.numel(), dtype=np.float32) * np.float32(0.02)
            elif id(p) in ln_gains:
                v = np.float32(1.0) + rng.standard_normal(p.numel(), dtype=np.float32) * np.float32(0.05)
            else:
                v = rng.standard_normal(p.numel(), dtype=np.float32) * np.float32(0.05)
            p.copy_(torch.from_numpy(v.reshape(tuple(p.shape))))
    return module
