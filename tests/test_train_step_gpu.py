"""One whole prompted-frame training step with nothing on the CPU -- frames -> SAM2Base.forward_image (HIP trunk + neck)
-> forward_first_frame (HIP prompt encoder + mask decoder) -> MultiStepMultiMasksAndIous (HIP loss sums) -> backward
through the heads, the neck and the LoRA blocks -- against autograd through the CPU restatements (oracle/hiera_oracle.py,
oracle/sam_heads.py, oracle/loss_oracle.py, each pinned to the reference's own outputs).  This is the single-frame,
no-memory slice of the reference's step (train_lora_sam.py:161-173): encoder LoRA factors (lora_qkv.py:148-192) and, when
wrapped, decoder LoRA factors (lora_qkv.py:112-143) are the only tensors with gradients.  fp32 operands; bar: the north
star's 1e-3 on the LoRA gradients."""
import pytest
import torch

from conftest import BPLUS, rel_err
from test_loss_cpu import CONFIGS
from test_trunk_gpu import make_encoder

pytestmark = pytest.mark.gpu


def _targets(n, size=1024):
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    t = torch.zeros(n, size, size)
    for i in range(n):
        cy, cx, r = 420 + 180 * i, 380 + 260 * i, 150 + 60 * i
        t[i] = (((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float()
    return t


@pytest.mark.timeout(1200)
def test_whole_training_step_vs_restatements():
    from image_segmentation_amd.loss_fns import CORE_LOSS_KEY, MultiStepMultiMasksAndIous
    from image_segmentation_amd.sam_heads import SAM2Base, wrap_decoder_lora
    from image_segmentation_amd.synth import synth_frames
    from oracle import hiera_oracle as O
    from oracle import loss_oracle as LO
    from oracle import sam_heads as SH
    from oracle.params import make_params
    n = 2
    coords = torch.tensor([[[400.0, 420.0]], [[640.0, 600.0]]])        # (x, y): inside each target disc
    labels = torch.ones(n, 1, dtype=torch.int32)
    targets = _targets(n)
    cfg = dict(CONFIGS["repo"])

    # ---- product ----------------------------------------------------------------------------------------------------------
    enc = make_encoder(BPLUS)
    enc.trunk.compute_dtype = torch.float32
    model = SAM2Base(image_encoder=enc)
    hp = SH.make_head_params(0)
    model.load_state_dict(hp, strict=False)
    model = model.cuda()
    for name, p in model.named_parameters():
        if not name.startswith("image_encoder."):
            p.requires_grad = False
    wrap_decoder_lora(model, 4, fresh_projections=False)
    gen = torch.Generator().manual_seed(5)
    dec_lora = []
    for blk in model.sam_mask_decoder.transformer.layers:
        sa = blk.self_attn
        for lin in (sa.A_q, sa.B_q, sa.A_v, sa.B_v):
            with torch.no_grad():
                lin.weight.copy_(0.05 * torch.randn(lin.weight.shape, generator=gen))
            dec_lora.append(lin.weight)
    assert all(w.requires_grad for w in dec_lora)
    out = model.forward_first_frame(model.forward_image(synth_frames(n, 1024, 0, device="cuda")), coords.cuda(), labels.cuda())
    crit = MultiStepMultiMasksAndIous(**{**cfg, "weight_dict": dict(cfg["weight_dict"])})
    outs = [{"multistep_pred_multimasks_high_res": [out["high_res_multimasks"]], "multistep_pred_ious": [out["ious"]],
             "multistep_object_score_logits": [out["object_score_logits"]]}]
    losses = crit(outs, targets.cuda().unsqueeze(0))
    losses[CORE_LOSS_KEY].backward()
    torch.cuda.synchronize()

    # ---- restatements, CPU autograd ------------------------------------------------------------------------------------------
    spec = O.block_specs(**BPLUS)
    p = make_params(spec, 4, 0)
    leaves = [w.detach().cpu().float().clone().requires_grad_() for w in dec_lora]
    merged = dict(hp)
    for l in range(2):
        aq, bq, av, bv = leaves[4 * l:4 * l + 4]
        pre = f"sam_mask_decoder.transformer.layers.{l}.self_attn."
        merged[pre + "q_proj.weight"] = hp[pre + "q_proj.weight"] + bq @ aq
        merged[pre + "v_proj.weight"] = hp[pre + "v_proj.weight"] + bv @ av
    fpn = O.image_encoder_forward(p, synth_frames(n, 1024, 0), spec)["backbone_fpn"]
    s0, s1 = SH.high_res_features(merged, fpn[0], fpn[1])
    pix = fpn[2] + merged["no_mem_embed"].view(1, SH.D, 1, 1)
    sparse = SH.embed_points(merged, coords, labels)
    dense = merged["sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(n, -1, SH.EMB, SH.EMB)
    masks, iou, obj = SH.predict_masks(merged, pix, SH.dense_pe(merged), sparse, dense, s0, s1)
    low = torch.where((obj > 0)[:, None, None], masks[:, 1:], torch.tensor(SH.NO_OBJ_SCORE))
    high = torch.nn.functional.interpolate(low, size=(1024, 1024), mode="bilinear", align_corners=False)
    ref = LO.multistep_losses([{"multistep_pred_multimasks_high_res": [high], "multistep_pred_ious": [iou[:, 1:]],
                                "multistep_object_score_logits": [obj]}], targets.unsqueeze(0), **cfg)
    ref[LO.CORE_LOSS_KEY].backward()

    errs = {k: abs(float(losses[k]) - float(ref[k])) / max(1e-6, abs(float(ref[k]))) for k in ("loss_mask", "loss_dice", "loss_iou", "loss_class", "core_loss")}
    n_enc = 0
    for name, prm in enc.named_parameters():
        if prm.requires_grad:
            errs["enc." + name.split("blocks.")[1]] = rel_err(prm.grad, p[name].grad)
            assert float(p[name].grad.norm()) > 0
            n_enc += 1
    for i, (w, r) in enumerate(zip(dec_lora, leaves)):
        errs[f"dec.lora{i}"] = rel_err(w.grad, r.grad)
    print("whole step, losses (product / restatement):", {k: (float(losses[k]), float(ref[k])) for k in ("loss_mask", "loss_dice", "loss_iou", "loss_class", "core_loss")})
    print("whole step vs restatements:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert float(ref["loss_mask"]) > 0 and float(ref["loss_dice"]) > 0 and float(ref["loss_class"]) > 0
    assert n_enc == 8
    assert all(v < 1e-3 for v in errs.values()), errs
