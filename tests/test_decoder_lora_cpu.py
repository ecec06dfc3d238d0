"""The heads oracle with decoder-side LoRA (oracle/sam_heads.py `_attention` + `decoder_lora_params`) against the outputs of
the REFERENCE's own `wrap_decoder_lora` / `Attention_LoRA` on its own SAM2Base (tests/golden/decoder_lora_r4.npz, written by
oracle/make_decoder_lora_goldens.py: lora_qkv.py:11-67,112-143): forward logits / IoU predictions / object scores, the fixed
functional's value, its gradient w.r.t. the image embedding and w.r.t. all 24 tensors the reference leaves trainable (8 LoRA
factors + the 16 fresh projection tensors).  CPU only; the GPU counterpart is tests/test_decoder_lora_gpu.py."""
import torch

from conftest import load_golden, rel_err


def oracle_decoder_lora_run(hp, fpn, coords, labels, wl, wi, wo, trainable):
    """Autograd through the restatement; `trainable` names become leaves.  Returns (outputs dict, loss, d_emb, grads dict)."""
    from oracle import sam_heads as SH
    hp = dict(hp)
    for n in trainable:
        hp[n] = hp[n].clone().requires_grad_()
    f2 = fpn[2].clone().requires_grad_()
    with torch.enable_grad():       # SH.forward_sam_heads detaches its inputs: the same steps, attached
        s0, s1 = SH.high_res_features(hp, fpn[0], fpn[1])
        pix = f2 + hp["no_mem_embed"].view(1, SH.D, 1, 1)
        sparse = SH.embed_points(hp, coords.float(), labels)
        dense = hp["sam_prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(f2.shape[0], -1, SH.EMB, SH.EMB)
        masks, iou, obj = SH.predict_masks(hp, pix, SH.dense_pe(hp), sparse, dense, s0, s1)
        low = torch.where((obj > 0)[:, None, None], masks[:, 1:], torch.tensor(SH.NO_OBJ_SCORE))
        loss = (low * wl).sum() + (iou[:, 1:] * wi).sum() + (obj * wo).sum()
        loss.backward()
    return dict(low=low.detach(), ious=iou[:, 1:].detach(), obj=obj.detach()), float(loss.detach()), f2.grad, {n: hp[n].grad for n in trainable}


def check_against_golden(g, out, loss, d_emb, grads, tol_fwd, tol_grad):
    assert rel_err(out["low"][:, :, ::2, ::2], g["low_multi"]) < tol_fwd
    assert abs(float(out["low"].double().norm()) - g["low_multi.stats"][2]) < tol_fwd * g["low_multi.stats"][2]
    assert rel_err(out["ious"], g["ious"]) < tol_fwd and rel_err(out["obj"], g["obj"]) < tol_fwd
    assert abs(loss - float(g["loss"])) < tol_grad * abs(float(g["loss"]))
    assert rel_err(d_emb[:, ::4, ::2, ::2], g["d_emb.sub"]) < tol_grad
    assert abs(float(d_emb.double().norm()) - float(g["d_emb.norm"])) < tol_grad * float(g["d_emb.norm"])
    names = sorted(k[len("grad."):] for k in g if k.startswith("grad.") and not k.startswith("grad.norm."))
    assert len(names) == 24 and sorted(grads) == names
    worst = 0.0
    scale = max(float(torch.as_tensor(g["grad." + n]).double().norm()) for n in names if n.endswith(".bias"))
    for n in names:
        gr = grads[n]
        if n.endswith("k_proj.bias"):
            # softmax(q (k + b)^T) does not depend on b (q . b is the same for every key), so this gradient is exactly zero in
            # exact arithmetic: the reference's own value is rounding noise, and so must ours be
            assert float(torch.as_tensor(g["grad." + n]).double().norm()) < 1e-5 * scale and float(gr.double().norm()) < 1e-4 * scale, n
            continue
        if "_proj.weight" in n:
            assert abs(float(gr.double().norm()) - float(g["grad.norm." + n])) < tol_grad * float(g["grad.norm." + n]), n
            gr = gr[::4, ::4]
        e = rel_err(gr, g["grad." + n])
        worst = max(worst, e)
        assert e < tol_grad, (n, e)
    return worst


def test_oracle_decoder_lora_vs_reference_golden():
    from oracle import sam_heads as SH
    g = load_golden("decoder_lora_r4")
    dl = SH.decoder_lora_params(int(g["rank"]), 0)
    hp = dict(SH.make_head_params(0))
    hp.update(dl)
    coords, labels = torch.from_numpy(g["coords"]), torch.from_numpy(g["labels"])
    B = coords.shape[0]
    wl, wi, wo = SH.functional_weights(B)
    out, loss, d_emb, grads = oracle_decoder_lora_run(hp, SH.synthetic_fpn(B), coords, labels, wl, wi, wo, sorted(dl))
    worst = check_against_golden(g, out, loss, d_emb, grads, 1e-5, 1e-4)
    print(f"oracle vs reference, decoder LoRA: worst gradient rel-err {worst:.2e}")
