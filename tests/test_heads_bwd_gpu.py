"""Backward of the SAM heads (image_segmentation_amd/heads_autograd.py, csrc/heads_bwd.hip): every kernel against torch
autograd of the same expression, then the whole heads -- gradient of a scalar of the masks / IoU predictions / object
scores with respect to the image embedding and to the decoder-side LoRA factors -- against autograd through the CPU
restatement oracle/sam_heads.py (pinned to the reference's masks in tests/test_masks_cpu.py).  fp32 operands: rel-err
(L2, whole tensor) <= 1e-3, the north star's bar for LoRA gradients (measured ~1e-6 .. 1.5e-4); bf16 operands: direction only."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return scale * torch.randn(*shape, generator=g)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("nq,nk,nh,hd", [(8, 8, 8, 32), (7, 4096, 8, 16), (4096, 9, 8, 16), (16, 300, 4, 32), (700, 16, 2, 32)])
def test_mha_bwd(dtype, tol, nq, nk, nh, hd):
    from image_segmentation_amd import ops
    B, C = 2, nh * hd
    q, k, v = (_rand(B, n, C, seed=s).to(dtype) for s, n in ((1, nq), (2, nk), (3, nk)))
    go = _rand(B, nq, C, seed=4)
    out = ops.mha(q.cuda(), k.cuda(), v.cuda(), nh)
    dq, dk, dv = ops.mha_bwd(q.cuda(), k.cuda(), v.cuda(), out, go.cuda(), nh)
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))

    def heads(x):
        return x.view(B, -1, nh, hd).transpose(1, 2)

    ref = F.scaled_dot_product_attention(heads(qr), heads(kr), heads(vr)).transpose(1, 2).reshape(B, nq, C)
    ref.backward(go.double())
    assert rel_err(dq, qr.grad) < tol and rel_err(dk, kr.grad) < tol and rel_err(dv, vr.grad) < tol


@pytest.mark.parametrize("C,ln", [(64, True), (32, False)])
def test_upscale2x2_bwd(C, ln):
    from image_segmentation_amd import ops
    B, H, W = 2, 6, 5
    g = _rand(B * H * W, 4 * C, seed=6)
    skip = _rand(B, 2 * H, 2 * W, C, seed=7)
    w, b = 1 + 0.1 * _rand(C, seed=8), 0.1 * _rand(C, seed=9)
    go = _rand(B, 2 * H, 2 * W, C, seed=10)
    dg = ops.upscale2x2_bwd(go.cuda(), g.cuda(), skip.cuda(), (w.cuda(), b.cuda()) if ln else None, 1e-6, B, H, W, C, torch.float32)
    gr = g.double().requires_grad_()
    x = gr.view(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C) + skip.double()
    if ln:
        x = F.layer_norm(x, (C,), w.double(), b.double(), 1e-6)
    F.gelu(x).backward(go.double())
    assert rel_err(dg, gr.grad) < 2e-5


def test_mask_gate_and_bilinear_bwd():
    from image_segmentation_amd import ops
    B, S = 3, 12
    obj = torch.tensor([[2.0], [-1.0], [0.5]])
    gl = _rand(B, 3, S * S, seed=1)
    dlog = ops.mask_gate_bwd(gl.cuda(), obj.cuda(), 1, 8).cpu()
    ref = torch.zeros(B, S * S, 8)
    ref[:, :, 1:4] = gl.permute(0, 2, 1)
    ref[1] = 0
    assert torch.equal(dlog, ref)
    for scale in (4, 2):
        x = _rand(B, 3, S, S, seed=2).double().requires_grad_()
        go = _rand(B, 3, S * scale, S * scale, seed=3)
        F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False).backward(go.double())
        got = ops.bilinear_upsample_bwd(go.cuda(), scale)
        assert rel_err(got, x.grad) < 1e-6


@pytest.mark.parametrize("alpha,gamma", [(0.25, 2.0), (-1.0, 0.0), (0.5, 1.5)])
def test_mask_loss_sums_and_bwd(alpha, gamma):
    from image_segmentation_amd import ops
    N, M, H = 3, 3, 40
    x = _rand(N, M, H, H, seed=5, scale=3.0)
    t = (_rand(N, H, H, seed=6) > 0.3).float()
    sums = ops.mask_loss_sums(x.cuda(), t.cuda(), alpha, gamma).cpu()
    xr = x.double().requires_grad_()
    tt = t.double().unsqueeze(1).expand_as(xr)
    p = xr.sigmoid()
    ce = F.binary_cross_entropy_with_logits(xr, tt, reduction="none")
    pt = p * tt + (1 - p) * (1 - tt)
    fl = ce * (1 - pt) ** gamma
    if alpha >= 0:
        fl = (alpha * tt + (1 - alpha) * (1 - tt)) * fl
    ref = torch.stack([fl.flatten(2).sum(-1), (p * tt).flatten(2).sum(-1), p.flatten(2).sum(-1), tt.flatten(2).sum(-1),
                       ((xr > 0) & (tt > 0)).flatten(2).sum(-1).double(), ((xr > 0) | (tt > 0)).flatten(2).sum(-1).double()], -1)
    assert rel_err(sums, ref.detach()) < 1e-5
    assert torch.equal(sums[..., 4:], ref[..., 4:].float())
    coef = _rand(N, M, 3, seed=7)
    coef[1, 2] = 0.0                                                   # an unselected candidate
    (ref[..., :3] * coef.double()).sum().backward()
    dx = ops.mask_loss_bwd(x.cuda(), t.cuda(), coef.cuda(), alpha, gamma)
    assert rel_err(dx, xr.grad) < 1e-5
    assert (dx[1, 2] == 0).all()


# ---- the whole heads ---------------------------------------------------------------------------------------------------
def _heads_with_lora(seed, dtype, rank=4):
    from image_segmentation_amd.sam_heads import wrap_decoder_lora
    from test_heads_gpu import _make_heads
    m, hp = _make_heads(seed, dtype)
    wrap_decoder_lora(m, rank, fresh_projections=False)                    # freezes the decoder, adds the factors
    for p in m.parameters():
        p.requires_grad = False
    gen = torch.Generator().manual_seed(21)
    lora = []
    for blk in m.sam_mask_decoder.transformer.layers:
        sa = blk.self_attn
        for lin in (sa.A_q, sa.B_q, sa.A_v, sa.B_v):
            with torch.no_grad():
                lin.weight.copy_(0.05 * torch.randn(lin.weight.shape, generator=gen))
            lin.weight.requires_grad = True
            lora.append(lin.weight)
    return m, hp, lora


def _oracle_loss_and_grads(hp, lora, fpn, coords, labels, wl, wi, wo):
    """Autograd through the restatement with the LoRA updates merged as W + B A (the factors are leaves)."""
    from oracle import sam_heads as SH
    f2 = fpn[2].clone().requires_grad_()
    leaves = [w.detach().cpu().float().clone().requires_grad_() for w in lora]
    merged = dict(hp)
    for l in range(2):
        aq, bq, av, bv = leaves[4 * l:4 * l + 4]
        pre = f"sam_mask_decoder.transformer.layers.{l}.self_attn."
        merged[pre + "q_proj.weight"] = hp[pre + "q_proj.weight"] + bq @ aq
        merged[pre + "v_proj.weight"] = hp[pre + "v_proj.weight"] + bv @ av
    with torch.enable_grad():       # SH.forward_sam_heads itself detaches its inputs: the same steps, attached
        feat_s0, feat_s1 = SH.high_res_features(merged, fpn[0], fpn[1])
        pix = f2 + merged["no_mem_embed"].view(1, SH.D, 1, 1)
        sparse = SH.embed_points(merged, coords.float(), labels)
        dense = merged["sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(f2.shape[0], -1, SH.EMB, SH.EMB)
        masks, iou, obj = SH.predict_masks(merged, pix, SH.dense_pe(merged), sparse, dense, feat_s0, feat_s1)
        low = torch.where((obj > 0)[:, None, None], masks[:, 1:], torch.tensor(SH.NO_OBJ_SCORE))
        loss = (low * wl).sum() + (iou[:, 1:] * wi).sum() + (obj * wo).sum()
        loss.backward()
    return float(loss), f2.grad, [w.grad for w in leaves]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_heads_backward_vs_restatement(dtype, tol):
    from oracle import sam_heads as SH
    from test_heads_gpu import _synthetic_fpn
    B = 2
    m, hp, lora = _heads_with_lora(0, dtype)
    fpn = _synthetic_fpn(B)
    coords = torch.tensor([[[300.0, 410.0]], [[700.0, 128.0]]])
    labels = torch.ones(B, 1, dtype=torch.int32)
    wl = _rand(B, 3, 256, 256, seed=31) / 256.0
    wi, wo = _rand(B, 3, seed=32), _rand(B, 1, seed=33)
    ref_loss, ref_df2, ref_dl = _oracle_loss_and_grads(hp, lora, fpn, coords, labels, wl, wi, wo)

    s0, s1 = SH.high_res_features(hp, fpn[0], fpn[1])
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
    f2 = cl(fpn[2]).requires_grad_()
    out = m.forward_first_frame({"backbone_fpn": [cl(s0), cl(s1), f2]}, coords.cuda(), labels.cuda())
    loss = (out["low_res_multimasks"] * wl.cuda()).sum() + (out["ious"] * wi.cuda()).sum() + (out["object_score_logits"] * wo.cuda()).sum()
    loss.backward()
    errs = {"loss": abs(float(loss) - ref_loss) / abs(ref_loss), "d_embedding": rel_err(f2.grad, ref_df2)}
    for i, (w, r) in enumerate(zip(lora, ref_dl)):
        errs[f"lora{i}"] = rel_err(w.grad, r)
    print(f"heads backward vs restatement ({dtype}): {errs}")
    assert errs["loss"] < tol
    if dtype == torch.float32:
        assert all(v < tol for v in errs.values()), errs
    else:
        # bf16 operands in the heads are offered but not parity-grade (tests/test_masks_gpu.py): this random-sign functional
        # cancels heavily, so only the direction of the embedding gradient is checked
        cos = F.cosine_similarity(f2.grad.float().flatten().cpu(), ref_df2.flatten(), dim=0).item()
        assert cos > 0.97 and all(torch.isfinite(w.grad).all() for w in lora), (cos, errs)
    assert float(ref_df2.norm()) > 0 and all(float(r.norm()) > 0 for r in ref_dl)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-5)])
def test_batched_outer(dtype, tol):
    from image_segmentation_amd import ops
    B, M, L, C = 3, 5000, 8, 32
    G, U = _rand(B, M, L, seed=1), _rand(B, M, C, seed=2).to(dtype)
    got = ops.batched_outer(G.cuda(), U.cuda())
    ref = torch.einsum("bml,bmc->blc", G.double(), U.double())
    assert rel_err(got, ref) < tol
