"""Mask goldens from the REFERENCE's own SAM2Base (TEST INFRASTRUCTURE ONLY; run once in the build
container:  python oracle/make_mask_goldens.py ).

Builds the reference model exactly as sam2_model.py:11-90 does (SAM2Base instead of its SAM2Train
subclass, which needs the absent `tensordict`), fills the image encoder from the repo's seeded
generator (as oracle/make_goldens.py) and the SAM heads from oracle/sam_heads.make_head_params, then
runs the first-frame click path of the reference:
  forward_image (sam2_base.py:467-479) -> + no_mem_embed (sam2_base.py:669-674)
  -> _forward_sam_heads(multimask_output=True) (sam2_base.py:257-413)
and stores the predicted masks as packed bits with the IoU / object-score heads' outputs.
No reference source is copied; only outputs are stored (tests/golden/masks_bplus_r4.npz)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402
import sam_heads as SH  # noqa: E402
from image_segmentation_amd.synth import synth_frames  # noqa: E402

CLICKS = [(400.0, 620.0), (700.0, 250.0)]   # (x, y) pixel coordinates, one positive click per frame


def build_reference_model(R):
    from SAM2.sam2.sam2.modeling.memory_attention import MemoryAttention, MemoryAttentionLayer
    from SAM2.sam2.sam2.modeling.memory_encoder import CXBlock, Fuser, MaskDownSampler, MemoryEncoder
    from SAM2.sam2.sam2.modeling.sam.transformer import RoPEAttention
    from SAM2.sam2.sam2.modeling.sam2_base import SAM2Base
    enc = MG.build_reference_encoder(R, MG.BPLUS, 4, 0)
    PES = R["PositionEmbeddingSine"]
    sa = RoPEAttention(rope_theta=10000.0, feat_sizes=[64, 64], embedding_dim=256, num_heads=1, downsample_rate=1, dropout=0.1)
    ca = RoPEAttention(rope_theta=10000.0, rope_k_repeat=True, feat_sizes=[64, 64], embedding_dim=256, num_heads=1,
                       downsample_rate=1, dropout=0.1, kv_in_dim=64)
    layer = MemoryAttentionLayer(activation="relu", dim_feedforward=2048, dropout=0.1, pos_enc_at_attn=False, d_model=256,
                                 pos_enc_at_cross_attn_keys=True, pos_enc_at_cross_attn_queries=False,
                                 self_attention=sa, cross_attention=ca)
    mem_att = MemoryAttention(d_model=256, pos_enc_at_input=True, layer=layer, num_layers=4)
    mem_enc = MemoryEncoder(out_dim=64, position_encoding=PES(num_pos_feats=64, normalize=True, temperature=10000),
                            mask_downsampler=MaskDownSampler(kernel_size=3, stride=2, padding=1),
                            fuser=Fuser(CXBlock(dim=256, kernel_size=7, padding=3, layer_scale_init_value=1e-6,
                                                use_dwconv=True), num_layers=2))
    model = SAM2Base(enc, memory_attention=mem_att, memory_encoder=mem_enc, num_maskmem=7, image_size=1024,
                     sigmoid_scale_for_mem_enc=20.0, sigmoid_bias_for_mem_enc=-10.0,
                     use_mask_input_as_output_without_sam=True, directly_add_no_mem_embed=True,
                     use_high_res_features_in_sam=True, multimask_output_in_sam=True, iou_prediction_use_sigmoid=True,
                     use_obj_ptrs_in_encoder=True, add_tpos_enc_to_obj_ptrs=True, only_obj_ptrs_in_the_past_for_eval=True,
                     pred_obj_scores=True, pred_obj_scores_mlp=True, fixed_no_obj_ptr=True,
                     multimask_output_for_tracking=True, use_multimask_token_for_obj_ptr=True, multimask_min_pt_num=0,
                     multimask_max_pt_num=1, use_mlp_for_obj_ptr_proj=True, no_obj_embed_spatial=True,
                     proj_tpos_enc_in_obj_ptrs=True, compile_image_encoder=False)
    hp = SH.make_head_params(0)
    sd = dict(model.named_parameters())
    sd.update(dict(model.named_buffers()))
    with torch.no_grad():
        for name, t in hp.items():
            assert tuple(sd[name].shape) == tuple(t.shape), name    # also pins head_param_shapes() to the reference
            sd[name].copy_(t)
    covered = set(hp)
    for name in sd:
        if name.startswith(("sam_prompt_encoder.", "sam_mask_decoder.")) and "mask_downscaling" not in name:
            assert name in covered, f"head tensor {name} not enumerated by oracle/sam_heads.py"
    return model.eval()


@torch.no_grad()
def reference_masks(model, frames, coords, labels):
    out = model.forward_image(frames)
    fpn = out["backbone_fpn"]
    B = frames.shape[0]
    # first (initial conditioning) frame: directly_add_no_mem_embed, sam2_base.py:669-674
    vis = fpn[-1].flatten(2).permute(2, 0, 1)                      # (HW)BC as _prepare_backbone_features
    pix = (vis + model.no_mem_embed).permute(1, 2, 0).view(B, 256, 64, 64)
    res = model._forward_sam_heads(backbone_features=pix, point_inputs={"point_coords": coords, "point_labels": labels},
                                   mask_inputs=None, high_res_features=[fpn[0], fpn[1]], multimask_output=True)
    low_multi, high_multi, ious, low, high, obj_ptr, obj_logits = res
    return dict(low_multi=low_multi, ious=ious, low=low, high=high, obj=obj_logits), out


def main():
    torch.set_num_threads(8)
    R = MG.import_reference()
    model = build_reference_model(R)
    n = len(CLICKS)
    frames = synth_frames(n, 1024, 0)
    coords = torch.tensor(CLICKS, dtype=torch.float32).view(n, 1, 2)
    labels = torch.ones(n, 1, dtype=torch.int32)
    ref, enc_out = reference_masks(model, frames, coords, labels)
    g = {"coords": coords.numpy(), "labels": labels.numpy(), "ious": ref["ious"].numpy(), "obj": ref["obj"].numpy(),
         "best": torch.argmax(ref["ious"], -1).numpy(),
         "low_bits": np.packbits((ref["low"] > 0).numpy().reshape(n, -1), axis=1),
         "high_bits": np.packbits((ref["high"] > 0).numpy().reshape(n, -1), axis=1),
         "low_multi.sub": ref["low_multi"][:, :, ::4, ::4].numpy().copy(),
         "low_multi.stats": np.array([ref["low_multi"].double().mean().item(), ref["low_multi"].double().std().item()])}
    path = os.path.join(MG.GOLD, "masks_bplus_r4.npz")
    np.savez_compressed(path, **g)
    area = (ref["high"] > 0).float().mean(dim=(1, 2, 3))
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB; ious {ref['ious'].tolist()} obj {ref['obj'].flatten().tolist()} "
          f"mask area {area.tolist()}")
    # self-check, printed only: the restatement on the reference encoder's raw outputs, and how a
    # bf16-sized relative error on the encoder outputs moves the masks
    hp = SH.make_head_params(0)
    with torch.no_grad():
        raw = model.image_encoder(frames)["backbone_fpn"]
    mine = SH.forward_sam_heads(hp, raw, coords, labels)
    err = float((mine["low_res_multimasks"] - ref["low_multi"]).norm() / ref["low_multi"].norm())
    print("restatement vs reference: low-res logits rel-err", err,
          "IoU", [SH.mask_iou(mine["high_res_masks"][i] > 0, ref["high"][i] > 0) for i in range(n)])
    for eps in (4e-3, 8e-3, 2e-2):
        gen = torch.Generator().manual_seed(1)
        noisy = [t + eps * t.std() * torch.randn(t.shape, generator=gen) for t in raw]
        pert = SH.forward_sam_heads(hp, noisy, coords, labels)
        print(f"feature noise {eps}: IoU", [round(SH.mask_iou(pert["high_res_masks"][i] > 0, ref["high"][i] > 0), 4) for i in range(n)],
              "same best:", bool((pert["best"] == torch.argmax(ref["ious"], -1)).all()))


if __name__ == "__main__":
    main()
