"""Decoder-side LoRA goldens from the REFERENCE's own `wrap_decoder_lora` / `Attention_LoRA` (lora_qkv.py:11-67,112-143)
on the reference's own SAM2Base -- TEST INFRASTRUCTURE ONLY; run once in the build container:

    python oracle/make_decoder_lora_goldens.py

The reference model is built as oracle/make_mask_goldens.py builds it (sam2_model.py:11-90 with SAM2Base in place of its
SAM2Train subclass), the heads are filled from oracle/sam_heads.make_head_params, then the reference's
`lora_qkv.wrap_decoder_lora(model, 4)` runs: it freezes the mask decoder and replaces every transformer layer's
self-attention by an `Attention_LoRA`, whose constructor creates FRESH, trainable q / k / v / out projections
(lora_qkv.py:26-29).  Those and the four factors per layer are then filled from oracle/sam_heads.decoder_lora_params
(non-zero B), and the first-click path of `_forward_sam_heads` (sam2_base.py:257-413) runs with autograd on a synthetic
FPN pyramid (oracle/sam_heads.synthetic_fpn: the encoder is not involved).  Stored: the low-resolution multimask logits,
IoU predictions and object score, the value of the fixed functional

    L = sum(low_res_multimasks * wl) + sum(ious * wi) + sum(object_score_logits * wo)

(wl, wi, wo = oracle/sam_heads.functional_weights, seeded), dL/d(image embedding) on a subsample, and dL/d(parameter) for EVERY parameter the reference
leaves trainable in the decoder after wrapping: the 8 LoRA factors in full and the 16 fresh projection tensors (weights
subsampled 4 x 4).  No reference source is copied; only inputs' seeds and outputs are stored
(tests/golden/decoder_lora_r4.npz)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402
import make_mask_goldens as MM  # noqa: E402
import sam_heads as SH  # noqa: E402

RANK = 4
COORDS = [[[300.0, 410.0]], [[700.0, 128.0]]]


def main():
    torch.set_num_threads(8)
    R = MG.import_reference()
    model = MM.build_reference_model(R)
    for p in model.parameters():                      # train_lora_sam.py:116-117
        p.requires_grad = False
    R["lora_qkv"].wrap_decoder_lora(model, RANK)      # the reference's own function
    dl = SH.decoder_lora_params(RANK, 0)
    sd = dict(model.named_parameters())
    with torch.no_grad():
        for name, t in dl.items():
            assert tuple(sd[name].shape) == tuple(t.shape), name
            sd[name].copy_(t)
    trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    # what the reference leaves trainable: the factors AND the fresh projections, nothing else in the model
    assert trainable == sorted(dl), (trainable, sorted(dl))
    B = len(COORDS)
    f0, f1, f2 = SH.synthetic_fpn(B)
    coords = torch.tensor(COORDS, dtype=torch.float32)
    labels = torch.ones(B, 1, dtype=torch.int32)
    wl, wi, wo = SH.functional_weights(B)
    md = model.sam_mask_decoder
    with torch.no_grad():                             # forward_image's conv_s0 / conv_s1 (sam2_base.py:473-478)
        s0, s1 = md.conv_s0(f0), md.conv_s1(f1)
    emb = f2.clone().requires_grad_()
    pix = emb + model.no_mem_embed.view(1, 256, 1, 1)         # sam2_base.py:669-674
    low_multi, high_multi, ious, low, high, obj_ptr, obj = model._forward_sam_heads(
        backbone_features=pix, point_inputs={"point_coords": coords, "point_labels": labels}, mask_inputs=None,
        high_res_features=[s0, s1], multimask_output=True)
    loss = (low_multi * wl).sum() + (ious * wi).sum() + (obj * wo).sum()
    loss.backward()
    g = {"coords": coords.numpy(), "labels": labels.numpy(), "rank": np.array(RANK),
         "low_multi": low_multi.detach().numpy()[:, :, ::2, ::2].copy(), "ious": ious.detach().numpy(), "obj": obj.detach().numpy(),
         "low_multi.stats": np.array([low_multi.double().mean().item(), low_multi.double().std().item(), low_multi.double().norm().item()]),
         "loss": np.array(loss.item()), "d_emb.sub": emb.grad.numpy()[:, ::4, ::2, ::2].copy(),
         "d_emb.norm": np.array(emb.grad.double().norm().item())}
    for name in trainable:
        gr = sd[name].grad
        assert gr is not None and float(gr.norm()) > 0, name
        if "_proj.weight" in name:
            g["grad.norm." + name] = np.array(gr.double().norm().item())
            gr = gr[::4, ::4]
        g["grad." + name] = gr.numpy().copy()
    path = os.path.join(MG.GOLD, "decoder_lora_r4.npz")
    np.savez_compressed(path, **g)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB; loss {loss.item():.6f}; ious {ious.tolist()}; obj {obj.flatten().tolist()}; "
          f"{len(trainable)} trainable tensors")
    # self-check, printed only: the restatement with the same parameters
    hp = dict(SH.make_head_params(0))
    hp.update(dl)
    ref = SH.forward_sam_heads(hp, (f0, f1, f2), coords, labels)
    print("restatement vs reference (decoder LoRA): low-res logits rel-err",
          float((ref["low_res_multimasks"] - low_multi.detach()).norm() / low_multi.detach().norm()))


if __name__ == "__main__":
    main()
