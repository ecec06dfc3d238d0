"""Decoder-side LoRA on the MI355X against the REFERENCE's own run (tests/golden/decoder_lora_r4.npz; lora_qkv.py:11-67,
112-143): `wrap_decoder_lora` with its default -- the reference's behaviour: fresh, TRAINABLE q / k / v / out projections per
wrapped layer plus the four factors -- forward logits to 1e-3 and the gradients of the fixed functional w.r.t. the image
embedding and all 24 trainable tensors to 1e-3 (fp32 operands: BASELINE.json's LoRA-gradient bar)."""
import pytest
import torch

from conftest import load_golden
from test_decoder_lora_cpu import check_against_golden

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_decoder_lora_vs_reference_golden():
    from image_segmentation_amd.sam_heads import Attention_LoRA, wrap_decoder_lora
    from oracle import sam_heads as SH
    from test_heads_gpu import _make_heads
    g = load_golden("decoder_lora_r4")
    rank = int(g["rank"])
    m, hp = _make_heads(0, torch.float32)
    for p in m.parameters():                      # train_lora_sam.py:116-117
        p.requires_grad = False
    wrap_decoder_lora(m, rank)                    # default: the reference's literal behaviour
    dl = SH.decoder_lora_params(rank, 0)
    sd = dict(m.named_parameters())
    trainable = sorted(n for n, p in sd.items() if p.requires_grad)
    assert trainable == sorted(dl)                # the same 24 tensors the reference leaves trainable
    assert all(isinstance(b.self_attn, Attention_LoRA) for b in m.sam_mask_decoder.transformer.layers)
    with torch.no_grad():
        for n, t in dl.items():
            sd[n].copy_(t)
    coords, labels = torch.from_numpy(g["coords"]), torch.from_numpy(g["labels"])
    B = coords.shape[0]
    fpn = SH.synthetic_fpn(B)
    wl, wi, wo = SH.functional_weights(B)
    s0, s1 = SH.high_res_features(hp, fpn[0], fpn[1])
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
    f2 = cl(fpn[2]).requires_grad_()
    out = m.forward_first_frame({"backbone_fpn": [cl(s0), cl(s1), f2]}, coords.cuda(), labels.cuda())
    loss = (out["low_res_multimasks"] * wl.cuda()).sum() + (out["ious"] * wi.cuda()).sum() + (out["object_score_logits"] * wo.cuda()).sum()
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: sd[n].grad.detach().cpu() for n in trainable}
    res = dict(low=out["low_res_multimasks"].detach().cpu(), ious=out["ious"].detach().cpu(), obj=out["object_score_logits"].detach().cpu())
    worst = check_against_golden(g, res, float(loss), f2.grad.detach().cpu(), grads, 1e-3, 1e-3)
    print(f"HIP heads vs reference, decoder LoRA with fresh trainable projections: worst gradient rel-err {worst:.2e}")
    # an optimizer step on all 24 tensors is seen by the next forward (packed weights are keyed by parameter version)
    opt = torch.optim.SGD([sd[n] for n in trainable], lr=0.1)
    opt.step()
    again = m.forward_first_frame({"backbone_fpn": [cl(s0), cl(s1), f2.detach()]}, coords.cuda(), labels.cuda())
    assert not torch.equal(again["low_res_multimasks"], out["low_res_multimasks"])
