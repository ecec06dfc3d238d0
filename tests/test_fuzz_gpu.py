"""Seeded random sweep over shapes the fixed cases of test_ops_gpu.py do not list: `lsam_linear` (ragged M, N, K, every
epilogue combination, leading dimensions wider than the rows) and `lsam_attention_fwd` (non-square grids, windows that hang
over the grid, q-pooling, head_dim 8 .. 128) against fp64 evaluations of the same expressions (the oracle's window
semantics via test_ops_gpu.ref_attention).  A geometry the library refuses (LsamError) is counted, never rerouted; the
sweep fails if it refuses more than a third.  Tolerances as in test_ops_gpu.py."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _t(gen, *shape):
    return torch.randn(*shape, generator=gen)


def test_linear_random_shapes():
    from image_segmentation_amd import ops
    from image_segmentation_amd._C import LsamError
    rng, gen = random.Random(1234), torch.Generator().manual_seed(1234)
    ran = refused = 0
    for case in range(70):
        dt = rng.choice([torch.float32, torch.bfloat16])
        M = rng.choice([1, 7, 16, 33, 130, 255, 256, 257, 1000, 1024, 1031, 2049, 5000, 9001])
        N = 8 * rng.randint(1, 90)
        K = 8 * rng.randint(1, 150)
        pad_a, pad_o = 8 * rng.randint(0, 3), 8 * rng.randint(0, 3)
        epi = rng.choice(["none", "bias", "res", "gelu", "gelu_pre", "relu", "sigmoid", "rowmod", "few_rows"])
        A = _t(gen, M, K + pad_a).to(dt).cuda()[:, :K]
        W = (_t(gen, N, K) / math.sqrt(K)).to(dt).cuda()
        b = _t(gen, N).cuda()
        ref = A.double() @ W.double().t()
        kw, want = {}, ref
        if epi != "none":
            kw["bias"] = b
            want = ref + b.double()
        if epi == "res":
            r = _t(gen, M, N).cuda()
            kw["res"], want = r, want + r.double()
        if epi == "rowmod" and M >= 16:
            r = _t(gen, 8, N).cuda()
            kw["res"], kw["res_row_mod"] = r, 8
            want = want + r.double().repeat((M + 7) // 8, 1)[:M]
        if epi in ("gelu", "gelu_pre"):
            kw["gelu"], want = True, F.gelu(want)
        if epi == "relu":
            kw["relu"], want = True, want.relu()
        if epi == "sigmoid":
            kw["sigmoid"], want = True, want.sigmoid()
        if epi == "few_rows":
            kw["few_rows"] = True
        out_f32 = rng.random() < 0.5 or dt == torch.float32
        wide = torch.full((M, N + pad_o), 3.0, dtype=torch.float32 if out_f32 else dt, device="cuda")
        try:
            if epi == "gelu_pre":
                got, pre = ops.linear(A, W, save_pre=True, out_f32=False, **kw)
                assert rel_err(pre.double(), ref + b.double()) < (1e-6 if dt == torch.float32 else 5e-3), (case, M, N, K, dt)
            else:
                got = ops.linear(A, W, out_f32=out_f32, out=wide[:, :N], **kw)
                assert bool((wide[:, N:] == 3.0).all()), ("wrote past column N", case, M, N, K, dt, epi)
        except LsamError:
            refused += 1
            continue
        ran += 1
        exact_in = got.dtype == torch.float32
        bar = 2e-6 if (dt == torch.float32 or (exact_in and epi not in ("gelu", "gelu_pre"))) else 6e-3
        if epi in ("gelu", "gelu_pre") and dt == torch.bfloat16:
            bar = 1e-2   # the bf16 kernels' GELU is the sigmoid form (within the bf16 budget)
        assert rel_err(got.double(), want) < bar, (case, M, N, K, dt, epi, out_f32)
    print(f"linear sweep: {ran} ran, {refused} refused")
    assert refused * 3 <= ran


def test_attention_random_geometries():
    from image_segmentation_amd import ops
    from image_segmentation_amd._C import LsamError
    from test_ops_gpu import ref_attention
    rng, gen = random.Random(99), torch.Generator().manual_seed(99)
    ran = refused = 0
    for case in range(48):
        dt = rng.choice([torch.float32, torch.bfloat16])
        ws = rng.choice([0, 2, 4, 6, 7, 8, 12, 14, 16])
        pool = rng.choice([1, 1, 2]) if ws > 0 and ws % 2 == 0 else 1
        H = 2 * rng.randint(2, 20)
        W = 2 * rng.randint(2, 20) if rng.random() < 0.6 else H
        if ws == 0:
            H, W = min(H, 24), min(W, 24)
        nh = rng.choice([1, 2, 3, 4, 8])
        hd = 8 * rng.choice([1, 3, 4, 7, 8, 9, 12, 16])
        N = rng.randint(1, 3)
        qkv = _t(gen, N, H, W, 3 * nh * hd).to(dt).cuda()
        bias = 0.5 * _t(gen, 3 * nh * hd)
        try:
            out, lse = ops.attention_fwd(qkv, bias.cuda(), ws, pool, nh, want_lse=True)
        except LsamError:
            refused += 1
            continue
        ran += 1
        bias_t = bias.cuda().to(dt).float().cpu()
        ref, lse_ref = ref_attention(qkv.float().cpu(), bias_t, H, W, ws, pool, nh)
        assert out.shape == ref.shape, (case, H, W, ws, pool, nh, hd)
        bar = 5e-6 if dt == torch.float32 else 8e-3
        assert rel_err(out.float(), ref) < bar, (case, dt, N, H, W, ws, pool, nh, hd)
        assert rel_err(lse, lse_ref) < (1e-5 if dt == torch.float32 else 3e-3), (case, dt, N, H, W, ws, pool, nh, hd)
    print(f"attention sweep: {ran} ran, {refused} refused")
    assert refused * 3 <= ran


def test_attention_bwd_random_geometries():
    """The backward the LoRA blocks use (windowed, optional 2x2 q-pool, head_dim <= 96) on random non-square grids."""
    from image_segmentation_amd import ops
    from image_segmentation_amd._C import LsamError
    from test_ops_gpu import ref_attention
    rng, gen = random.Random(7), torch.Generator().manual_seed(7)
    ran = refused = 0
    for case in range(28):
        dt = rng.choice([torch.float32, torch.bfloat16])
        ws = rng.choice([4, 6, 7, 8, 14, 16])
        pool = 2 if ws > 8 else (rng.choice([1, 1, 2]) if ws % 2 == 0 else 1)   # the backward takes <= 64 queries per window
        H, W = 2 * rng.randint(2, 14), 2 * rng.randint(2, 14)
        nh, hd = rng.choice([1, 2, 4]), 8 * rng.choice([2, 4, 7, 9, 12])
        N = rng.randint(1, 2)
        qkv = _t(gen, N, H, W, 3 * nh * hd).to(dt).cuda()
        bias = (0.5 * _t(gen, 3 * nh * hd)).cuda()
        do = _t(gen, N, H // pool, W // pool, nh * hd).to(dt).cuda()
        try:
            _, lse = ops.attention_fwd(qkv, bias, ws, pool, nh, want_lse=True)
            dqkv = ops.attention_bwd(qkv, bias, do, lse, ws, nh, pool=pool)
        except LsamError as e:
            print("refused:", (H, W, ws, pool, nh, hd), str(e)[:120])
            refused += 1
            continue
        ran += 1
        x = qkv.float().cpu().double().requires_grad_(True)
        ref, _ = ref_attention(x, bias.to(dt).float().cpu(), H, W, ws, pool, nh)
        (ref * do.float().cpu().double()).sum().backward()
        assert rel_err(dqkv.float(), x.grad) < (2e-5 if dt == torch.float32 else 1e-2), (case, dt, N, H, W, ws, pool, nh, hd)
    print(f"attention backward sweep: {ran} ran, {refused} refused")
    assert refused * 3 <= ran


def test_heads_mha_random_shapes():
    """lsam_mha_fwd / lsam_mha_bwd (the SAM heads' attention: one side has few rows) on random lengths."""
    from image_segmentation_amd import ops
    from image_segmentation_amd._C import LsamError
    rng, gen = random.Random(5), torch.Generator().manual_seed(5)
    ran = refused = 0
    for case in range(24):
        dt = rng.choice([torch.float32, torch.bfloat16])
        nh, hd = rng.choice([1, 2, 4, 8]), rng.choice([16, 32])
        few, many = rng.randint(1, 16), rng.choice([1, 5, 16, 100, 777, 4096])
        nq, nk = (few, many) if rng.random() < 0.5 else (many, few)
        B, C = rng.randint(1, 3), nh * hd
        q, k, v = (_t(gen, B, n, C).to(dt) for n in (nq, nk, nk))
        go = _t(gen, B, nq, C)
        try:
            out = ops.mha(q.cuda(), k.cuda(), v.cuda(), nh)
            dq, dk, dv = ops.mha_bwd(q.cuda(), k.cuda(), v.cuda(), out, go.cuda(), nh)
        except LsamError:
            refused += 1
            continue
        ran += 1
        qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))
        heads = lambda x: x.view(B, -1, nh, hd).transpose(1, 2)
        ref = F.scaled_dot_product_attention(heads(qr), heads(kr), heads(vr)).transpose(1, 2).reshape(B, nq, C)
        ref.backward(go.double())
        tol = 2e-5 if dt == torch.float32 else 2e-2
        assert rel_err(out.double(), ref.detach()) < (2e-6 if dt == torch.float32 else 6e-3), (case, dt, B, nq, nk, nh, hd)
        for got, want in ((dq, qr.grad), (dk, kr.grad), (dv, vr.grad)):
            if float(want.norm()) < 1e-9 * float(go.norm()):   # one key: dq and dk vanish analytically
                assert float(got.double().norm()) < 1e-5 * float(go.norm()), (case, dt, B, nq, nk, nh, hd)
            else:
                assert rel_err(got, want) < tol, (case, dt, B, nq, nk, nh, hd)
    print(f"mha sweep: {ran} ran, {refused} refused")
    assert refused * 3 <= ran


def _random_hiera_config(rng):
    embed, heads = rng.choice([(32, 1), (48, 1), (56, 1), (64, 2), (96, 1), (112, 2)])
    stages = tuple(rng.randint(1, 3) for _ in range(4))
    ends = [sum(stages[:i]) - 1 for i in range(1, 5)]
    # a global block must not be a q-pool block (the reference itself divides by zero there)
    cand = [i for i in range(ends[1] + 2, ends[2] + 1)]
    glob = (rng.choice(cand),) if cand and rng.random() < 0.7 else ()
    return dict(embed_dim=embed, num_heads=heads, stages=stages, global_att_blocks=glob,
                window_spec=rng.choice([(8, 4, 14, 7), (8, 4, 16, 8), (4, 4, 8, 4), (8, 8, 8, 8)]))


@pytest.mark.timeout(900)
def test_random_hiera_configs_vs_oracle():
    """Whole encoder (trunk + neck) forward and LoRA gradients, fp32, on Hiera configurations no golden covers: random stage
    depths (so the two LoRA blocks land on same-width blocks, on a stage-entry block, or straddle a stage boundary), widths,
    window specs, an optional global block, image sizes and ranks -- against the CPU oracle on the same seeded tensors.
    Bars: FPN levels 1e-4, LoRA gradients 1e-3 (BASELINE.json)."""
    from image_segmentation_amd._C import LsamError
    from image_segmentation_amd.synth import synth_frames
    from oracle import hiera_oracle as O
    from oracle.params import make_params
    from test_trunk_gpu import fpn_loss, make_encoder
    rng = random.Random(2024)
    ran = refused = 0
    for case in range(8):
        cfg = _random_hiera_config(rng)
        size, rank, n = rng.choice([128, 192, 256]), rng.choice([2, 4, 8]), rng.randint(1, 2)
        img = synth_frames(n, size, case)
        try:
            enc = make_encoder(cfg, rank=rank, seed=case)
            enc.trunk.compute_dtype = torch.float32
            out = enc(img.cuda())
            loss = fpn_loss(out)
            loss.backward()
        except (LsamError, NotImplementedError) as e:
            print("refused:", cfg, size, str(e)[:100])
            refused += 1
            continue
        ran += 1
        spec = O.block_specs(**cfg)
        p = make_params(spec, rank, case)
        ref = O.image_encoder_forward(p, img, spec)
        O.fpn_loss(ref).backward()
        for lvl, (a, b) in enumerate(zip(out["backbone_fpn"], ref["backbone_fpn"])):
            assert a.shape == b.shape and rel_err(a, b.detach()) < 1e-4, (case, cfg, size, lvl)
        got = {k: v.grad for k, v in enc.named_parameters() if v.requires_grad}
        assert len(got) == 8
        for name, g in got.items():
            assert rel_err(g, p[name].grad) < 1e-3, (case, cfg, size, rank, name)
        # the same configuration on bf16 operands (the bench's path): embedding bar 2e-2
        enc.trunk.compute_dtype = torch.bfloat16
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out16 = enc(img.cuda())
        err16 = rel_err(out16["backbone_fpn"][2].float(), ref["backbone_fpn"][2].detach())
        assert err16 < 2e-2, (case, cfg, size, err16)
    print(f"hiera config sweep: {ran} ran, {refused} refused")
    assert ran >= 5
