"""SAM heads on the MI355X: the kernels of csrc/heads.hip one by one against plain torch, the
product `SAM2Base` (image_segmentation_amd/sam_heads.py) against the CPU restatement oracle/sam_heads.py (pinned to the
reference's own masks in tests/test_masks_cpu.py), and `wrap_decoder_lora` / `Attention_LoRA` against the same
restatement with the rank-r updates merged into the dense weights.  Tolerances: fp32 compute 2e-4 on the token-level
outputs and 1e-3 on the mask logits (relative to the output's largest value; the ~40 accumulation stages differ in order
from torch's), bf16 compute by mask IoU >= 0.99 (the north-star bar)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(*shape, generator=g)).cuda()


# ---- kernels ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("adt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_add_rows(adt, odt):
    from image_segmentation_amd import ops
    a = _rand(3 * 50, 256, seed=1).to(adt)
    b = _rand(50, 256, seed=2)
    out = ops.add_rows(a, b, odt)
    ref = (a.float().view(3, 50, 256) + b).view(-1, 256).to(odt)
    assert out.dtype == odt and torch.equal(out, ref)
    assert torch.equal(ops.add_rows(a, None, odt), a.to(odt))
    one = ops.add_rows(a, b[:1], odt)
    assert torch.equal(one, (a.float() + b[:1]).to(odt))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("nq,nk,nh,hd", [(8, 8, 8, 32), (7, 4096, 8, 16), (4096, 9, 8, 16), (130, 300, 4, 64), (3, 70, 8, 32)])
def test_mha(dtype, tol, nq, nk, nh, hd):
    from image_segmentation_amd import ops
    B, C = 2, nh * hd
    q, k, v = (_rand(B, n, C, seed=s).to(dtype) for s, n in ((1, nq), (2, nk), (3, nk)))
    out = ops.mha(q, k, v, nh)

    def heads(x):
        return x.float().view(B, -1, nh, hd).transpose(1, 2)

    ref = F.scaled_dot_product_attention(heads(q), heads(k), heads(v)).transpose(1, 2).reshape(B, nq, C)
    assert out.dtype == dtype and rel_err(out, ref) < tol


def test_pe_encode():
    from image_segmentation_amd import ops
    G = _rand(2, 128, seed=4)
    c = torch.rand(500, 2, generator=torch.Generator().manual_seed(5)).cuda()
    out = ops.pe_encode(G, c)
    x = 2 * math.pi * ((2 * c.double() - 1) @ G.double())
    ref = torch.cat([x.sin(), x.cos()], -1).float()
    assert (out - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("C,ln", [(64, True), (32, False)])
def test_upscale2x2(dtype, tol, C, ln):
    from image_segmentation_amd import ops
    B, H, W = 2, 6, 5
    g = _rand(B * H * W, 4 * C, seed=6)
    skip = _rand(B, 2 * H, 2 * W, C, seed=7).to(dtype)
    w, b = 1 + 0.1 * _rand(C, seed=8), 0.1 * _rand(C, seed=9)
    out = ops.upscale2x2(g, skip, (w, b) if ln else None, 1e-6, B, H, W, C, dtype)
    x = g.view(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C) + skip.float()
    if ln:
        x = F.layer_norm(x, (C,), w, b, 1e-6)
    ref = F.gelu(x)
    assert out.dtype == dtype and rel_err(out, ref) < tol
    out0 = ops.upscale2x2(g, None, None, 0.0, B, H, W, C, dtype)
    ref0 = F.gelu(g.view(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C))
    assert rel_err(out0, ref0) < tol


def test_mask_gate_and_bilinear():
    from image_segmentation_amd import ops
    B, S = 3, 24
    logits = _rand(B, S * S, 8, seed=10)
    obj = torch.tensor([[2.0], [-1.0], [0.0]]).cuda()
    low = ops.mask_gate(logits, obj, 1, 3, -1024.0)
    ref = logits[:, :, 1:4].permute(0, 2, 1).clone()
    ref[1:] = -1024.0
    assert torch.equal(low, ref)
    assert torch.equal(ops.mask_gate(logits, None, 0, 1, -1024.0), logits[:, :, :1].permute(0, 2, 1))
    x = low.view(B, 3, S, S).contiguous()
    for scale in (4, 2):
        up = ops.bilinear_upsample(x, scale)
        refu = F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False)
        assert (up - refu).abs().max().item() < 1e-6 * refu.abs().max().item() + 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_linear_relu_sigmoid_epilogues(dtype, tol):
    from image_segmentation_amd import ops
    for M, N, K in ((8, 2048, 256), (5, 8, 256), (4096, 128, 256)):
        a, w, b = _rand(M, K, seed=11).to(dtype), (_rand(N, K, seed=12) / math.sqrt(K)).to(dtype), _rand(N, seed=13)
        pre = a.float() @ w.float().t() + b
        assert rel_err(ops.linear(a, w, b, relu=True, out_f32=True), F.relu(pre)) < tol
        assert rel_err(ops.linear(a, w, b, sigmoid=True, out_f32=True), torch.sigmoid(pre)) < tol
        assert rel_err(ops.linear(a, w, b, relu=True).float(), F.relu(pre)) < max(tol, 4e-3 if dtype == torch.bfloat16 else 0)


# ---- the heads against the restatement -------------------------------------------------------------------------------
def _make_heads(seed=0, dtype=torch.float32):
    from image_segmentation_amd.sam_heads import SAM2Base
    from oracle import sam_heads as SH
    m = SAM2Base(image_encoder=None)
    hp = SH.make_head_params(seed)
    missing, unexpected = m.load_state_dict(hp, strict=True)        # the reference's state_dict keys, all 139 of them
    assert not missing and not unexpected
    m.compute_dtype = dtype
    return m.cuda(), hp


def _synthetic_fpn(B, seed=3):
    from oracle import sam_heads as SH
    return SH.synthetic_fpn(B, seed)


def _run_product(m, hp, fpn, coords, labels):
    """Feed the product heads what forward_image would: conv_s0 / conv_s1 of levels 0 / 1 (computed here by the oracle's
    own function, fp32 CPU), channels-last."""
    from oracle import sam_heads as SH
    s0, s1 = SH.high_res_features(hp, fpn[0], fpn[1])
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
    return m.forward_first_frame({"backbone_fpn": [cl(s0), cl(s1), cl(fpn[2])]}, coords.cuda(), labels.cuda())


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_heads_vs_restatement(dtype):
    from oracle import sam_heads as SH
    B = 2
    m, hp = _make_heads(0, dtype)
    fpn = _synthetic_fpn(B)
    coords = torch.tensor([[[300.0, 410.0]], [[700.0, 128.0]]])
    labels = torch.ones(B, 1, dtype=torch.int32)
    ref = SH.forward_sam_heads(hp, fpn, coords, labels)
    out = _run_product(m, hp, fpn, coords, labels)
    errs = {k: rel_err(out[k], ref[k]) for k in ("ious", "object_score_logits", "low_res_multimasks", "high_res_multimasks")}
    iou_lo = [SH.mask_iou(out["low_res_masks"][i].cpu() > 0, ref["low_res_masks"][i] > 0) for i in range(B)]
    iou_hi = [SH.mask_iou(out["high_res_masks"][i].cpu() > 0, ref["high_res_masks"][i] > 0) for i in range(B)]
    print(f"heads vs restatement ({dtype}): {errs}, IoU low {iou_lo} high {iou_hi}")
    assert (out["best"].cpu() == ref["best"]).all()
    if dtype == torch.float32:
        assert errs["ious"] < 2e-4 and errs["object_score_logits"] < 2e-4
        assert errs["low_res_multimasks"] < 1e-3 and errs["high_res_multimasks"] < 1e-3
        assert min(iou_lo + iou_hi) > 0.999
    else:
        assert errs["ious"] < 3e-2 and errs["object_score_logits"] < 3e-2
        assert min(iou_lo + iou_hi) >= 0.99


def test_heads_prompt_forms_and_gating():
    """Two points per frame with mixed labels, the no-prompt form, single-mask output, and the NO_OBJ_SCORE gate."""
    from oracle import sam_heads as SH
    B = 2
    m, hp = _make_heads(1)
    fpn = _synthetic_fpn(B, seed=5)
    coords = torch.tensor([[[100.0, 900.0], [512.0, 512.0]], [[20.0, 30.0], [1000.0, 640.0]]])
    labels = torch.tensor([[1, 0], [1, -1]], dtype=torch.int32)
    ref = SH.forward_sam_heads(hp, fpn, coords, labels)
    out = _run_product(m, hp, fpn, coords, labels)
    assert rel_err(out["low_res_multimasks"], ref["low_res_multimasks"]) < 1e-3
    assert rel_err(out["ious"], ref["ious"]) < 2e-4
    # sparse embeddings alone (prompt_encoder.py:75-111)
    emb = m._embed_points(coords.cuda(), labels.cuda())
    assert (emb.cpu() - SH.embed_points(hp, coords, labels)).abs().max().item() < 1e-4
    assert (m.get_dense_pe("cuda").cpu().t().reshape(1, 256, 64, 64) - SH.dense_pe(hp)).abs().max().item() < 1e-4
    # object gate: a negative object score replaces every mask with NO_OBJ_SCORE (sam2_base.py:359-368)
    with torch.no_grad():
        m.sam_mask_decoder.pred_obj_score_head.layers[2].bias.fill_(-4.0)
    gated = _run_product(m, hp, fpn, coords, labels)
    assert (gated["object_score_logits"] < 0).all()
    assert (gated["low_res_multimasks"] == -1024.0).all() and (gated["high_res_masks"] == -1024.0).all()
    with torch.no_grad():
        m.sam_mask_decoder.pred_obj_score_head.layers[2].bias.fill_(4.0)
    # single-mask form and the no-prompt form of the reference's own entry point
    s0, s1 = SH.high_res_features(hp, fpn[0], fpn[1])
    pix = (fpn[2] + hp["no_mem_embed"].view(1, 256, 1, 1)).cuda()
    low, high, ious, low_b, high_b, ptr_, obj = m._forward_sam_heads(
        pix, {"point_coords": coords.cuda(), "point_labels": labels.cuda()}, None, [s0.cuda(), s1.cuda()], multimask_output=False)
    assert low.shape == (B, 1, 256, 256) and high.shape == (B, 1, 1024, 1024) and ious.shape == (B, 1) and ptr_ is None
    masks, iou, _ = SH.predict_masks(hp, pix.cpu(), SH.dense_pe(hp), SH.embed_points(hp, coords, labels),
                                     hp["sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, 64, 64), s0, s1)
    assert rel_err(low, masks[:, :1]) < 1e-3 and rel_err(ious, iou[:, :1]) < 2e-4
    assert torch.equal(low, low_b) and torch.equal(high, high_b)
    none = m._forward_sam_heads(pix, None, None, [s0.cuda(), s1.cuda()], multimask_output=True)
    pad = SH.predict_masks(hp, pix.cpu(), SH.dense_pe(hp),
                           SH.embed_points(hp, torch.zeros(B, 0, 2), torch.zeros(B, 0, dtype=torch.int32)),
                           hp["sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, 64, 64), s0, s1)
    # the reference's no-prompt form is one padding point at (0, 0) with label -1 plus the usual pad (sam2_base.py:301-304)
    two_pads = SH.predict_masks(hp, pix.cpu(), SH.dense_pe(hp),
                                SH.embed_points(hp, torch.zeros(B, 1, 2), -torch.ones(B, 1, dtype=torch.int32)),
                                hp["sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, 64, 64), s0, s1)
    assert rel_err(none[0], two_pads[0][:, 1:]) < 1e-3
    assert pad[0].shape == two_pads[0].shape


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decoder_lora(dtype):
    """wrap_decoder_lora (lora_qkv.py:112-143): freezes the decoder, B = 0 leaves the output unchanged; with B != 0 the
    output equals the restatement run with W_q + B_q A_q and W_v + B_v A_v merged into the dense weights."""
    from image_segmentation_amd.sam_heads import Attention_LoRA, wrap_decoder_lora
    from oracle import sam_heads as SH
    B = 1
    m, hp = _make_heads(2, dtype)
    fpn = _synthetic_fpn(B, seed=7)
    coords, labels = torch.tensor([[[600.0, 333.0]]]), torch.ones(B, 1, dtype=torch.int32)
    base = _run_product(m, hp, fpn, coords, labels)
    wrap_decoder_lora(m, 4, fresh_projections=False)
    layers = m.sam_mask_decoder.transformer.layers
    assert all(isinstance(b.self_attn, Attention_LoRA) for b in layers)
    assert not any(p.requires_grad for n, p in m.sam_mask_decoder.named_parameters() if ".A_" not in n and ".B_" not in n)
    assert all(p.requires_grad for n, p in m.sam_mask_decoder.named_parameters() if ".A_" in n or ".B_" in n)
    same = _run_product(m, hp, fpn, coords, labels)
    assert torch.equal(same["low_res_multimasks"], base["low_res_multimasks"])          # B = 0
    merged = dict(hp)
    gen = torch.Generator().manual_seed(11)
    scale = 0.05 if dtype == torch.float32 else 4.0        # bf16: the update must clear the bf16 noise floor (~1e-2) to be seen
    for l, blk in enumerate(layers):
        sa = blk.self_attn
        with torch.no_grad():
            sa.B_q.weight.copy_(scale * torch.randn(sa.B_q.weight.shape, generator=gen))
            sa.B_v.weight.copy_(scale * torch.randn(sa.B_v.weight.shape, generator=gen))
        pre = f"sam_mask_decoder.transformer.layers.{l}.self_attn."
        merged[pre + "q_proj.weight"] = hp[pre + "q_proj.weight"] + (sa.B_q.weight @ sa.A_q.weight).detach().cpu().float()
        merged[pre + "v_proj.weight"] = hp[pre + "v_proj.weight"] + (sa.B_v.weight @ sa.A_v.weight).detach().cpu().float()
    ref = SH.forward_sam_heads(merged, fpn, coords, labels)
    out = _run_product(m, hp, fpn, coords, labels)
    moved = rel_err(out["low_res_multimasks"], base["low_res_multimasks"])
    err = rel_err(out["low_res_multimasks"], ref["low_res_multimasks"])
    print(f"decoder LoRA ({dtype}): update moved the logits by {moved:.3e}, error vs merged restatement {err:.3e}")
    assert moved > (10 if dtype == torch.float32 else 3) * err
    assert err < (1e-3 if dtype == torch.float32 else 5e-2)
    if dtype == torch.float32:
        assert SH.mask_iou(out["low_res_masks"][0].cpu() > 0, ref["low_res_masks"][0] > 0) >= 0.999
    # the reference's literal constructor (fresh projections) is available and produces different weights
    m2, _ = _make_heads(2, dtype)
    wrap_decoder_lora(m2, 4, fresh_projections=True)
    assert not torch.equal(m2.sam_mask_decoder.transformer.layers[0].self_attn.q_proj.weight.cpu(),
                           hp["sam_mask_decoder.transformer.layers.0.self_attn.q_proj.weight"])


def test_heads_fail_loudly_without_gpu_tensors():
    from image_segmentation_amd._C import LsamError
    m, hp = _make_heads(0)
    with pytest.raises(LsamError):
        m._forward_sam_heads(torch.zeros(1, 256, 64, 64), None, None, [torch.zeros(1, 32, 256, 256), torch.zeros(1, 64, 128, 128)])
    with pytest.raises(LsamError):
        m.forward_image(torch.zeros(1, 3, 1024, 1024).cuda())


def test_decoder_lora_training_steps_refresh_packed_weights():
    """The heads keep packed copies of their weights keyed by (data_ptr, _version): optimizer steps on the decoder LoRA
    factors must be seen by the next forward, `load_state_dict` drops every packed image, and an edit through `.data`
    (which does not bump the version) is picked up after `invalidate_packed()`."""
    from image_segmentation_amd.sam_heads import wrap_decoder_lora
    B = 1
    fpn = _synthetic_fpn(B, seed=5)
    coords, labels = torch.tensor([[[400.0, 700.0]]]), torch.ones(B, 1, dtype=torch.int32)

    def fresh_copy(src):
        m2, hp2 = _make_heads(4)
        wrap_decoder_lora(m2, 4, fresh_projections=False)
        m2.load_state_dict(src.state_dict(), strict=True)
        return m2, hp2

    m, hp = _make_heads(4)
    wrap_decoder_lora(m, 4, fresh_projections=False)
    for n, p in m.named_parameters():          # train_lora_sam.py:116-117 freezes everything but the LoRA factors
        p.requires_grad = ".A_" in n or ".B_" in n
    lora = [p for p in m.parameters() if p.requires_grad]
    assert len(lora) == 8
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in lora:
            p.copy_(0.05 * torch.randn(p.shape, generator=gen))
    opt = torch.optim.SGD(lora, lr=0.5)
    outs = []
    for _ in range(3):
        out = _run_product(m, hp, fpn, coords, labels)
        outs.append(out["low_res_multimasks"].detach().clone())
        (out["low_res_multimasks"] ** 2).mean().backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])       # every step was seen
    after = _run_product(m, hp, fpn, coords, labels)["low_res_multimasks"].detach()
    m2, hp2 = fresh_copy(m)                                                               # same parameters, nothing cached
    assert torch.equal(after, _run_product(m2, hp2, fpn, coords, labels)["low_res_multimasks"].detach())
    # load_state_dict into a model that has already packed: the old images must not survive
    m3, hp3 = _make_heads(4)
    wrap_decoder_lora(m3, 4, fresh_projections=False)
    _run_product(m3, hp3, fpn, coords, labels)
    m3.load_state_dict(m.state_dict(), strict=True)
    assert torch.equal(after, _run_product(m3, hp3, fpn, coords, labels)["low_res_multimasks"].detach())
    # an edit through .data is invisible until invalidate_packed()
    w = m3.sam_mask_decoder.transformer.layers[0].self_attn.B_q.weight
    w.data.mul_(-1.0)
    stale = _run_product(m3, hp3, fpn, coords, labels)["low_res_multimasks"].detach()
    m3.invalidate_packed()
    seen = _run_product(m3, hp3, fpn, coords, labels)["low_res_multimasks"].detach()
    assert torch.equal(stale, after) and not torch.equal(seen, after)
