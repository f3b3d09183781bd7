"""Diagnostics a caller can run on the tensors CodeFormer.forward returns (plain torch ops on the outputs: nothing here is on the hot path)."""
import torch


def top2_gap(logits):
    """Per-token gap between the largest and the second largest logit of a `(B, T, codebook)` tensor, and its minimum.

    The code index of a token is `argmax(logits)` (reference: `topk(softmax(logits), 1)`, codeformer_arch.py:257-258).  The HIP path reproduces
    the reference's logits to ~1e-5 (7e-6 on the goldens, tolerance 1e-4), so an index can only differ from the reference's where this gap is of
    that order.  The gate that admitted Winograd F(4x4,3x3) into the ENCODER (tests/test_gpu_real_images.py::test_encoder_logit_margin)
    was measured with seed-0 random-init weights -- the reference ships no checkpoint -- on seeded noise, the reference's real crops and
    synthetic range variants: smallest (gap / 2 x logit error) 7.0.  With a trained checkpoint, look at this number on your own crops: if the
    smallest gap comes near 1e-4, set CODEFORMER_HIP_F43_ENCODER=0 (F(2x2,3x3) in the encoder: a fifth of the per-layer error, -4 % / -9 % speed).
    Returns (gaps (B, T), min gap as a Python float)."""
    top = torch.topk(logits.detach().float(), 2, dim=-1).values
    gaps = top[..., 0] - top[..., 1]
    return gaps, float(gaps.min())
