"""Native FPN neck (FpnNeck.forward, image_encoder.py:102-134; SAM2Base.forward_image, sam2_base.py:467-479): one HIP
kernel per level, the top-down adjoint + one GEMM in the backward, conv_s0 / conv_s1 composed into the lateral convs."""
import types

import pytest
import torch
import torch.nn.functional as F

from conftest import TINY, load_golden, rel_err
from test_trunk_gpu import fpn_loss, make_encoder

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.bfloat16, 6e-3)])
@pytest.mark.parametrize("K,co", [(112, 256), (224, 256), (448, 256), (896, 256), (1152, 256), (56, 256), (112, 32), (224, 64)])
@pytest.mark.parametrize("with_top", [False, True])
def test_neck_level_kernel(dt, tol, K, co, with_top):
    from image_segmentation_amd import ops
    N, H, W = 2, 10, 6          # 120 tokens: ragged against every token tile
    x = rnd(N, H, W, K, seed=K)
    w = rnd(co, K, seed=co) / K ** 0.5
    b = rnd(co, seed=3)
    top = rnd(N, H // 2, W // 2, co, seed=4).to(dt) if with_top else None
    img = ops.neck_conv_pack(w.cuda(), dt)
    got = ops.neck_level_fwd(x.cuda(), img, b.cuda(), dt, co, top.cuda() if with_top else None)
    xr, wr = (x.to(dt).double(), w.to(dt).double()) if dt == torch.bfloat16 else (x.double(), w.double())
    ref = xr @ wr.T + b.double()
    if with_top:
        ref = ref + F.interpolate(top.double().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    assert got.dtype == dt and tuple(got.shape) == (N, H, W, co)
    assert rel_err(got.float(), ref) < tol


@pytest.mark.parametrize("K", [448, 896, 1152, 56])
@pytest.mark.parametrize("N,H,W", [(2, 10, 6), (9, 64, 64)])         # ragged small grid; 36864 tokens (the 64-token tile variant)
def test_neck_level_split_kernel(K, N, H, W):
    """Two-term bf16 operands: the product is fp32-accurate (no operand rounding), the bf16 output is rounded once."""
    from image_segmentation_amd import ops
    co = 256
    x = rnd(N, H, W, K, seed=K)
    w = rnd(co, K, seed=co) / K ** 0.5
    b = rnd(co, seed=3)
    top = rnd(N, H // 2, W // 2, co, seed=4)
    img = ops.neck_conv_pack_split(w.cuda())
    ref = x.double() @ w.double().T + b.double()
    out, out32 = ops.neck_level_fwd_split(x.cuda(), img, b.cuda(), co, None, want32=True)
    assert out.dtype == torch.bfloat16 and out32.dtype == torch.float32 and tuple(out.shape) == (N, H, W, co)
    assert rel_err(out32, ref) < 2e-5                      # 2^-16-class: the dropped lo.lo term and the bf16 lo roundings
    assert torch.equal(out, out32.bfloat16())              # one rounding
    ref = ref + F.interpolate(top.double().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    out, none = ops.neck_level_fwd_split(x.cuda(), img, b.cuda(), co, top.cuda(), want32=False)
    assert none is None
    assert rel_err(out.float(), ref) < 2.5e-3              # bf16 output rounding alone (plain bf16 operands: 6e-3 bound above)
    assert rel_err(out.float(), ref.float().bfloat16().float()) < 3e-4   # and almost always the SAME bf16 value as exact


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_neck_topdown_adjoint(dt):
    from image_segmentation_amd import ops
    g = rnd(2, 8, 12, 256, seed=1).to(dt)
    gc = rnd(2, 4, 6, 256, seed=2).to(dt)
    t = torch.zeros(2, 256, 4, 6, dtype=torch.float64, requires_grad=True)
    (F.interpolate(t, scale_factor=2.0, mode="nearest") * g.double().permute(0, 3, 1, 2)).sum().backward()
    want = t.grad.permute(0, 2, 3, 1)
    assert rel_err(ops.neck_topdown_bwd(g.cuda()).float(), want) < (1e-6 if dt == torch.float32 else 4e-3)
    assert rel_err(ops.neck_topdown_bwd(g.cuda(), gc.cuda()).float(), want + gc.double()) < (1e-6 if dt == torch.float32 else 4e-3)


def test_no_torch_arithmetic_kernels_in_encoder_forward():
    """The encoder forward launches only liblsam kernels (plus memcpy-class copies): profile one bf16 forward."""
    from image_segmentation_amd.synth import synth_frames
    enc = make_encoder(TINY)
    enc.trunk.compute_dtype = torch.bfloat16
    img = synth_frames(2, 256, 0, device="cuda")
    with torch.no_grad():
        enc(img)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            enc(img)
            torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
    foreign = [n for n in names if "lsam" not in n and "Memcpy" not in n and "Memset" not in n and "copyBuffer" not in n]
    assert names and not foreign, foreign


def _holder(enc, seed=0):
    from oracle.sam_heads import synth_head_tensor
    md = "sam_mask_decoder."
    dec = types.SimpleNamespace(conv_s0=torch.nn.Conv2d(256, 32, 1), conv_s1=torch.nn.Conv2d(256, 64, 1))
    with torch.no_grad():
        for nm in ("conv_s0", "conv_s1"):
            c = getattr(dec, nm)
            c.weight.copy_(synth_head_tensor(md + nm + ".weight", tuple(c.weight.shape), seed))
            c.bias.copy_(synth_head_tensor(md + nm + ".bias", tuple(c.bias.shape), seed))
            c.requires_grad_(False)
            c.cuda()
    return types.SimpleNamespace(image_encoder=enc, use_high_res_features_in_sam=True, sam_mask_decoder=dec)


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 1e-4, 1e-3), (torch.bfloat16, 2e-2, 6e-2)])
def test_forward_image_composed_high_res_convs(dtype, tol, gtol):
    """forward_image = ImageEncoder.forward + conv_s0 / conv_s1 on levels 0 / 1 (sam2_base.py:473-478), against the
    reference's FPN goldens pushed through the same two convolutions, and the LoRA gradients of the golden run (levels
    0 and 1 do not depend on the LoRA blocks, so the golden gradients of sum_l mean(fpn_l^2) are those of level 2)."""
    import image_segmentation_amd as m
    from image_segmentation_amd.synth import synth_frames
    from oracle import hiera_oracle as O
    from oracle.params import make_params
    g = load_golden("tiny_r4")
    enc = make_encoder(TINY)
    enc.trunk.compute_dtype = dtype
    model = _holder(enc)
    img = synth_frames(2, 256, 0, device="cuda")
    out = m.forward_image(model, img)
    f0, f1, f2 = out["backbone_fpn"]
    assert tuple(f0.shape) == (2, 32, 64, 64) and tuple(f1.shape) == (2, 64, 32, 32) and tuple(f2.shape) == (2, 256, 16, 16)
    assert [tuple(p.shape) for p in out["vision_pos_enc"]] == [(2, 256, 64, 64), (2, 256, 32, 32), (2, 256, 16, 16)]
    assert rel_err(f2.float(), g["fpn2"]) < tol
    spec = O.block_specs(**TINY)
    p = make_params(spec, 4, 0)
    with torch.no_grad():
        ref = O.image_encoder_forward(p, synth_frames(2, 256, 0), spec)
        dec = model.sam_mask_decoder
        r0 = F.conv2d(ref["backbone_fpn"][0], dec.conv_s0.weight.cpu(), dec.conv_s0.bias.cpu())
        r1 = F.conv2d(ref["backbone_fpn"][1], dec.conv_s1.weight.cpu(), dec.conv_s1.bias.cpu())
    assert rel_err(f0.float(), r0) < tol and rel_err(f1.float(), r1) < tol
    (f2.float() ** 2).mean().backward()
    torch.cuda.synchronize()
    n = 0
    for name, prm in enc.named_parameters():
        if prm.requires_grad:
            assert rel_err(prm.grad, g["grad." + name]) < gtol, name
            n += 1
    assert n == 8
    # without high-res features the function is the encoder itself
    model.use_high_res_features_in_sam = False
    with torch.no_grad():
        plain = m.forward_image(model, img)
    assert tuple(plain["backbone_fpn"][0].shape) == (2, 256, 64, 64)
