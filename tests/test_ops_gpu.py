"""GPU parity of every C-ABI operator against the CPU oracle / plain torch fp32 math on the
same seeded inputs.  fp32 kernels: rel-err <= 1e-5 per op (the end-to-end budget is 1e-4);
bf16 kernels: <= 1e-2 per op (end-to-end budget 2e-2)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dt, f32=1e-5, bf16=1e-2):
    return f32 if dt == torch.float32 else bf16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(scope="module")
def ops():
    from image_segmentation_amd import ops as o
    return o


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 336, 112), (200, 112, 448), (128, 256, 896), (77, 32, 224),
                                   (512, 2688, 928), (130, 56, 160),
                                   # >= 1024 rows and N % 112 == 0: the LDS-DMA 256x112-tile kernel, incl. ragged M / K tails
                                   (1101, 336, 112), (2048, 448, 448), (1500, 112, 1792), (1024, 2688, 928),
                                   (4099, 224, 224), (1280, 112, 160)])
def test_linear_bias_res(ops, dt, M, N, K):
    A, W, b, res = rnd(M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3), rnd(M, N, seed=4)
    Ad, Wd = A.cuda().to(dt), W.cuda().to(dt)
    ref = Ad.float().cpu().double() @ Wd.float().cpu().double().t() + b.double() + res.double()
    out = ops.linear(Ad, Wd, b.cuda(), res=res.cuda(), out_f32=True)
    assert rel_err(out, ref) < tol(dt, 1e-6, 1e-6)   # inputs identical, fp32 accumulate either way
    out_t = ops.linear(Ad, Wd, b.cuda())
    assert out_t.dtype == dt
    assert rel_err(out_t.float(), ref - res.double()) < tol(dt, 1e-6, 5e-3)


@pytest.mark.parametrize("M,N,K", [(33000, 672, 224), (20000, 1344, 480), (70000, 224, 64), (66000, 448, 32),
                                   (40000, 448, 448), (140000, 224, 896)])
def test_linear_persistent_many_tiles(ops, M, N, K):
    """bf16, N % 224 == 0, more output tiles than workgroups: every workgroup of the persistent kernel walks several
    tiles (pipelined k-tiles across tile boundaries, bias slices by tile parity, ragged last row tile, K tails that
    end in the first k-step of a k-tile, single-k-tile products)."""
    dt = torch.bfloat16
    A, W = rnd(M, K, seed=1).cuda().to(dt), (rnd(N, K, seed=2) / math.sqrt(K)).cuda().to(dt)
    b, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    ref = A.double() @ W.double().t() + b.double()
    out = ops.linear(A, W, b, res=res, out_f32=True)
    assert rel_err(out.double(), ref + res.double()) < 1e-6
    out = ops.linear(A, W, b)
    assert rel_err(out.double(), ref) < 5e-3
    out = ops.linear(A, W, None, out_f32=True)
    assert rel_err(out.double(), ref - b.double()) < 1e-6
    out, pre = ops.linear(A, W, b, gelu=True, save_pre=True)
    assert rel_err(pre.double(), ref) < 5e-3 and rel_err(out.double(), F.gelu(ref)) < 5e-3


@pytest.mark.parametrize("M,N,K", [(33000, 432, 144), (20000, 576, 144), (5000, 288, 288), (70000, 144, 576), (9000, 1152, 288),
                                   (1100, 3456, 1184), (2000, 152, 64), (40000, 864, 320), (1500, 96 * 3, 96),
                                   # N % 192 == 0: the 192-wide tile variant (no guard columns), incl. half k-tiles and ragged M
                                   (33000, 576, 224), (5000, 1728, 576), (2049, 192, 96), (70000, 384, 480), (66000, 768, 32)])
def test_linear_column_guarded_tiles(ops, M, N, K):
    """bf16, N NOT a multiple of 224 (Hiera-L widths are multiples of 144, Hiera-T/S of 96): the 256 x 224 persistent kernel
    with zero-filled weight rows and column-guarded epilogues -- every epilogue flavour, ragged M, K tails, and a poisoned
    output buffer to show that nothing is written past column N."""
    dt = torch.bfloat16
    A, W = rnd(M, K, seed=1).cuda().to(dt), (rnd(N, K, seed=2) / math.sqrt(K)).cuda().to(dt)
    b, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    ref = A.double() @ W.double().t() + b.double()
    wide = torch.full((M, N + 40), 7.0, device="cuda")
    ops.linear(A, W, b, res=res, out_f32=True, out=wide[:, :N])
    assert rel_err(wide[:, :N].double(), ref + res.double()) < 1e-6
    assert bool((wide[:, N:] == 7.0).all())
    wide16 = torch.full((M, N + 40), 7.0, device="cuda", dtype=dt)
    ops.linear(A, W, b, out=wide16[:, :N])
    assert rel_err(wide16[:, :N].double(), ref) < 5e-3
    assert bool((wide16[:, N:] == 7.0).all())
    out, pre = ops.linear(A, W, b, gelu=True, save_pre=True)
    assert rel_err(pre.double(), ref) < 5e-3 and rel_err(out.double(), F.gelu(ref)) < 5e-3
    aux = rnd(M, N, seed=6).cuda().to(dt)
    out = ops.linear(A, W, None, gelu_bwd_aux=aux, out_f32=True)
    x = aux.float().cpu().double().requires_grad_(True)
    F.gelu(x).sum().backward()
    assert rel_err(out, (ref.cpu() - b.cpu().double()) * x.grad) < 2e-5


def test_linear_operand_over_2GB_is_row_sliced(ops):
    """An A operand of 2 GB or more (Hiera-L at 64 frames) exceeds the 32-bit buffer offsets of the LDS-DMA kernels: the
    launcher cuts it into row slices.  Checked on row samples from the first slice, the seam and the ragged end."""
    dt = torch.bfloat16
    M, N, K = 1_100_037, 224, 1024                     # A: 2.25 GB
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn(M, K, device="cuda", generator=g, dtype=torch.float32).to(dt)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dt)
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    out = ops.linear(A, W, b, res=res, out_f32=True)
    seam = int(1.9e9 / (K * 2)) // 256 * 256
    for lo in (0, seam - 500, M - 700):
        sl = slice(lo, lo + 700)
        ref = A[sl].double() @ W.double().t() + b.double() + res[sl].double()
        assert rel_err(out[sl].double(), ref) < 1e-6, lo
    out16, pre = ops.linear(A, W, b, gelu=True, save_pre=True)
    sl = slice(seam - 300, seam + 300)
    ref = A[sl].double() @ W.double().t() + b.double()
    assert rel_err(pre[sl].double(), ref) < 5e-3 and rel_err(out16[sl].double(), F.gelu(ref)) < 5e-3


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M", [192, 1216])
def test_linear_epilogues(ops, dt, M):
    N, K = 448, 112
    A, W, b = rnd(M, K, seed=1).cuda().to(dt), (rnd(N, K, seed=2) / math.sqrt(K)).cuda().to(dt), rnd(N, seed=3).cuda()
    pre_ref = A.float().cpu().double() @ W.float().cpu().double().t() + b.cpu().double()
    out, pre = ops.linear(A, W, b, gelu=True, save_pre=True)
    assert rel_err(pre.float(), pre_ref) < tol(dt, 1e-6, 5e-3)
    assert rel_err(out.float(), F.gelu(pre_ref)) < tol(dt, 1e-6, 5e-3)
    # row-periodic residual (positional embedding)
    pos = rnd(64, N, seed=5).cuda()
    out = ops.linear(A, W, b, res=pos, res_row_mod=64, out_f32=True)
    assert rel_err(out, pre_ref + pos.cpu().double().repeat(M // 64, 1)) < 1e-6
    # GELU backward epilogue: out = acc * gelu'(aux)
    aux = rnd(M, N, seed=6).cuda().to(dt)
    out = ops.linear(A, W, None, gelu_bwd_aux=aux, out_f32=True)
    x = aux.float().cpu().double().requires_grad_(True)
    F.gelu(x).sum().backward()
    assert rel_err(out, (pre_ref - b.cpu().double()) * x.grad) < 2e-6
    # in-place residual
    y = rnd(M, N, seed=7).cuda()
    y0 = y.clone()
    ops.linear(A, W, b, res=y, out_f32=True, out=y)
    assert rel_err(y, pre_ref + y0.cpu().double()) < 1e-6


def test_linear_rejects_bad_shapes(ops):
    from image_segmentation_amd._C import LsamError
    A, W = torch.zeros(16, 20, device="cuda"), torch.zeros(12, 20, device="cuda")
    with pytest.raises(LsamError):
        ops.linear(A, W)  # N % 8 != 0
    with pytest.raises(LsamError):
        ops.linear(torch.zeros(4, 4), torch.zeros(8, 4))  # CPU tensors: no fallback


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C", [56, 112, 224, 448, 896, 1152])
def test_layernorm_fwd_bwd(ops, dt, C):
    M = 300
    x = rnd(M, C, seed=1) * 2 + 0.5
    g, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    y, st = ops.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), dt, want_stats=True)
    xr = x.double().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), g.double(), b.double(), eps=1e-6)
    assert rel_err(y.float(), ref.detach()) < tol(dt, 2e-6, 5e-3)
    assert rel_err(st[:, 0], x.double().mean(1)) < 1e-5
    dy, dres = rnd(M, C, seed=4), rnd(M, C, seed=5)
    ref.backward(dy.double())
    dx, dxt = ops.layernorm_bwd(dy.cuda(), x.cuda(), st, g.cuda(), dres.cuda(), dt)
    assert rel_err(dx, xr.grad + dres.double()) < 1e-5
    assert rel_err(dxt.float(), xr.grad + dres.double()) < tol(dt, 1e-5, 5e-3)
    # strided output (the LoRA-augmented buffer)
    y2 = ops.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), dt, ld=C + 32)
    assert torch.equal(y2[:, :C], y) and float(y2[:, C:].abs().max()) == 0.0


def ref_attention(qkv, bias, H, W, ws, pool, nh):
    """Oracle windows/pad/pool semantics on an explicit qkv tensor (fp64)."""
    from oracle import hiera_oracle as O
    N = qkv.shape[0]
    dout = qkv.shape[-1] // 3
    x = qkv.double()
    if ws > 0:
        # pad tokens carry qkv == bias: partition (qkv - bias) with zero padding, then add bias back
        xw, pad_hw = O.window_partition(x - bias.double(), ws)
        xw = xw + bias.double()
    else:
        xw = x
    B, h, w, _ = xw.shape
    q, k, v = xw.reshape(B, h * w, 3, nh, -1).unbind(2)
    if pool == 2:
        q = O.max_pool_nhwc(q.reshape(B, h, w, -1), (2, 2))
        h, w = q.shape[1:3]
        q = q.reshape(B, h * w, nh, -1)
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    lse = torch.logsumexp(s, dim=-1)                       # [B, nh, nq]
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, h, w, dout)
    lse = lse.transpose(1, 2).reshape(B, h, w, nh)
    Ho, Wo = H // pool, W // pool
    if ws > 0:
        wq = ws // pool
        php = (math.ceil(Ho / wq) * wq, math.ceil(Wo / wq) * wq)
        o = O.window_unpartition(o, wq, php, (Ho, Wo))
        lse = O.window_unpartition(lse, wq, php, (Ho, Wo))
    return o, lse


ATT_CASES = [  # (H, ws, pool, nh)  -- Hiera-B+ block shapes scaled down, incl. every pad/pool combination
    (16, 8, 1, 2), (16, 8, 2, 2), (16, 4, 1, 4), (8, 4, 2, 4), (16, 14, 1, 2), (16, 14, 2, 4),
    (18, 7, 1, 4), (32, 7, 1, 16), (16, 0, 1, 2), (64, 0, 1, 1), (64, 14, 1, 2),
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H,ws,pool,nh", ATT_CASES)
def test_attention_fwd(ops, dt, H, ws, pool, nh):
    N, hd = 2, 56
    qkv = rnd(N, H, H, 3 * nh * hd, seed=H + ws).cuda().to(dt)
    bias = rnd(3 * nh * hd, seed=9) * 0.5
    out, lse = ops.attention_fwd(qkv, bias.cuda(), ws, pool, nh, want_lse=True)
    bias_t = bias.cuda().to(dt).float().cpu()   # pad tokens hold the bias rounded to T
    ref, lse_ref = ref_attention(qkv.float().cpu(), bias_t, H, H, ws, pool, nh)
    assert out.shape == ref.shape
    assert rel_err(out.float(), ref) < tol(dt, 3e-6, 6e-3)
    assert rel_err(lse, lse_ref) < tol(dt, 1e-5, 2e-3)


@pytest.mark.parametrize("N,H,W,nh,hd", [(1, 16, 16, 2, 56), (2, 64, 64, 2, 56), (3, 64, 64, 8, 56), (2, 40, 64, 4, 56), (5, 30, 18, 8, 56),
                                         (2, 28, 28, 3, 64), (2, 42, 42, 16, 32), (1, 64, 64, 8, 56)])
def test_attention_fwd_window14_inference_form(ops, N, H, W, nh, hd):
    """The inference form (no log-sum-exp) of the 14 x 14 windows in bf16: more (window, head) items than the chip has
    workgroup slots, fewer, one; windows cut by the right / bottom edge (pad-token keys = bias), non-square grids, head_dim 64
    (no free padding channel for the softmax denominator) and 32.  Against the fp32 reference, and bit for bit against the
    form that also returns the log-sum-exp (the LoRA blocks' forward)."""
    ws, dt = 14, torch.bfloat16
    qkv = rnd(N, H, W, 3 * nh * hd, seed=N + H + W).cuda().to(dt)
    bias = rnd(3 * nh * hd, seed=9) * 0.5
    out = ops.attention_fwd(qkv, bias.cuda(), ws, 1, nh)
    ref, _ = ref_attention(qkv.float().cpu(), bias.cuda().to(dt).float().cpu(), H, W, ws, 1, nh)
    assert out.shape == ref.shape
    assert rel_err(out.float(), ref) < 6e-3
    old, _ = ops.attention_fwd(qkv, bias.cuda(), ws, 1, nh, want_lse=True)
    assert torch.equal(out, old)
    again = ops.attention_fwd(qkv, bias.cuda(), ws, 1, nh)
    assert torch.equal(out, again)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hd", [72, 96])
@pytest.mark.parametrize("H,ws,pool,nh", [(16, 8, 1, 2), (16, 8, 2, 2), (32, 16, 1, 4), (32, 16, 2, 2), (16, 0, 1, 2), (64, 0, 1, 1),
                                          (40, 16, 1, 2), (28, 14, 1, 2), (28, 14, 2, 2)])
def test_attention_fwd_head_dim_72(ops, dt, H, ws, pool, nh, hd):
    """Hiera-L heads (head_dim 72, windows 8/16, global) and Hiera-T/S heads (96): the streaming LDS-DMA kernel on
    256-byte tile rows (bf16, >= 64 keys and >= 32 queries per window) and the generic kernel otherwise."""
    N = 2
    qkv = rnd(N, H, H, 3 * nh * hd, seed=H + ws + 1).cuda().to(dt)
    bias = rnd(3 * nh * hd, seed=10) * 0.5
    out, lse = ops.attention_fwd(qkv, bias.cuda(), ws, pool, nh, want_lse=True)
    ref, lse_ref = ref_attention(qkv.float().cpu(), bias.cuda().to(dt).float().cpu(), H, H, ws, pool, nh)
    assert rel_err(out.float(), ref) < tol(dt, 3e-6, 6e-3)
    assert rel_err(lse, lse_ref) < tol(dt, 1e-5, 2e-3)
    out = ops.attention_fwd(qkv, bias.cuda(), ws, pool, nh)      # inference form (no log-sum-exp): the LDS-DMA kernel in bf16
    assert rel_err(out.float(), ref) < tol(dt, 3e-6, 6e-3)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H,ws,nh", [(32, 7, 16), (18, 7, 4), (16, 8, 2), (8, 4, 2)])
def test_attention_bwd(ops, dt, H, ws, nh):
    N, hd = 2, 56
    qkv = rnd(N, H, H, 3 * nh * hd, seed=3).cuda().to(dt)
    bias = (rnd(3 * nh * hd, seed=9) * 0.5).cuda()
    do = rnd(N, H, H, nh * hd, seed=4).cuda().to(dt)
    _, lse = ops.attention_fwd(qkv, bias, ws, 1, nh, want_lse=True)
    dqkv = ops.attention_bwd(qkv, bias, do, lse, ws, nh)
    x = qkv.float().cpu().double().requires_grad_(True)
    ref, _ = ref_attention(x, bias.to(dt).float().cpu(), H, H, ws, 1, nh)
    (ref * do.float().cpu().double()).sum().backward()
    assert rel_err(dqkv.float(), x.grad) < tol(dt, 2e-5, 1e-2)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H,ws,pool,nh,hd", [
    (16, 8, 1, 2, 72), (18, 7, 1, 2, 96),          # Hiera-L (72) and Hiera-T/S (96) heads, same-width LoRA blocks
    (16, 14, 2, 2, 56), (32, 14, 2, 4, 96),        # stage-entry block: 49 pooled queries x 196 keys (4 key tiles), padded grid
    (32, 16, 2, 2, 72), (16, 8, 2, 2, 56), (8, 4, 2, 1, 56),
])
def test_attention_bwd_general(ops, dt, H, ws, pool, nh, hd):
    """The tiled backward: head_dim up to 96, any number of keys, 2x2 q-pool with the gradient routed to the argmax token."""
    N = 2
    # values rounded to multiples of 1/8 would tie inside 2x2 groups; keep them generic and compare against autograd of
    # the oracle's max-pool, whose tie rule (first maximum) the kernel follows
    qkv = rnd(N, H, H, 3 * nh * hd, seed=3 + hd).cuda().to(dt)
    bias = (rnd(3 * nh * hd, seed=9) * 0.5).cuda()
    do = rnd(N, H // pool, H // pool, nh * hd, seed=4).cuda().to(dt)
    _, lse = ops.attention_fwd(qkv, bias, ws, pool, nh, want_lse=True)
    dqkv = ops.attention_bwd(qkv, bias, do, lse, ws, nh, pool=pool)
    x = qkv.float().cpu().double().requires_grad_(True)
    ref, _ = ref_attention(x, bias.to(dt).float().cpu(), H, H, ws, pool, nh)
    (ref * do.float().cpu().double()).sum().backward()
    assert rel_err(dqkv.float(), x.grad) < tol(dt, 2e-5, 1e-2)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H,ws,nh", [(16, 8, 2), (32, 16, 4), (32, 4, 4), (28, 14, 8), (64, 14, 2)])
def test_attention_fwd_with_pooled_queries(ops, dt, H, ws, nh):
    """do_pool(q) (hieradet.py:69-71) handed in instead of being taken from qkv: every forward kernel, bit for bit;
    the q columns of qkv are poisoned to show that they are not read."""
    hd, N = 56, 2
    d = nh * hd
    qkv = rnd(N, H, H, 3 * d, seed=31).cuda().to(dt)
    bias = (0.1 * rnd(3 * d, seed=32)).cuda()
    want, lse_want = ops.attention_fwd(qkv, bias, ws, 2, nh, want_lse=True)
    qp = F.max_pool2d(qkv[..., :d].float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous()
    poisoned = qkv.clone()
    poisoned[..., :d] = float("nan")
    got, lse = ops.attention_fwd(poisoned, bias, ws, 2, nh, want_lse=True, q_pooled=qp)
    if ws == 4 and dt == torch.bfloat16:   # 4 x 4 windows with pooled queries run the one-wave-per-window kernel: same math, other order
        assert rel_err(got.float(), want.float()) < 6e-3 and rel_err(lse, lse_want) < 1e-5
    else:
        assert torch.equal(got, want) and torch.equal(lse, lse_want)


def test_maxpool_cast_im2col(ops):
    x = rnd(2, 8, 12, 24, seed=1)
    ref = F.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.equal(ops.maxpool2x2(x.cuda()).cpu(), ref)
    a = rnd(33, 40, seed=2)
    assert torch.equal(ops.cast(a.cuda(), torch.bfloat16).cpu(), a.to(torch.bfloat16))
    img = rnd(2, 3, 32, 32, seed=3)
    cols = ops.im2col_patch(img.cuda(), torch.float32).cpu()
    ref = F.unfold(img, kernel_size=7, padding=3, stride=4).transpose(1, 2).reshape(-1, 147)
    assert torch.equal(cols[:, :147], ref) and float(cols[:, 147:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES)
def test_outer_reduce(ops, dt):
    M, C, R = 1000, 448, 4
    X, Y = rnd(M, C + 32, seed=1).cuda().to(dt), rnd(M, 48, seed=2).cuda().to(dt)
    ref = X[:, :C].float().cpu().double().t() @ Y[:, 16:16 + R].float().cpu().double()
    g = ops.outer_reduce(X[:, :C], Y[:, 16:32], R, transpose_out=False)
    assert rel_err(g, ref) < 1e-5
    gt = ops.outer_reduce(X[:, :C], Y[:, 16:32], R, transpose_out=True)
    assert rel_err(gt, ref.t()) < 1e-5
    g2 = ops.outer_reduce(X[:, :C], Y[:, 16:32], R, transpose_out=False)
    assert torch.equal(g, g2)   # deterministic two-pass reduction


def _win_block_ref(x, g1, b1, wqkv, bqkv, wproj, bproj, nh):
    """fp32 torch restatement of the attention half of MultiScaleBlock.forward (hieradet.py:134-163)
    for a same-width block with 8x8 windows and no padding."""
    N, H, W, d = x.shape
    xn = F.layer_norm(x, (d,), g1, b1, 1e-6)
    xw = xn.view(N, H // 8, 8, W // 8, 8, d).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64, d)
    qkv = (xw @ wqkv.t() + bqkv).reshape(-1, 64, 3, nh, d // nh)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(-1, 64, d)
    o = o @ wproj.t() + bproj
    o = o.view(N, H // 8, W // 8, 8, 8, d).permute(0, 1, 3, 2, 4, 5).reshape(N, H, W, d)
    return x + o


@pytest.mark.parametrize("nh,hd,N,H", [(2, 56, 2, 32), (1, 56, 3, 16), (2, 64, 1, 24), (1, 32, 2, 8)])
def test_fused_window_block(ops, nh, hd, N, H):
    d = nh * hd
    x = rnd(N, H, H, d, seed=1)
    g1, b1 = 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3)
    wqkv, bqkv = rnd(3 * d, d, seed=4) / math.sqrt(d), 0.2 * rnd(3 * d, seed=5)
    wproj, bproj = rnd(d, d, seed=6) / math.sqrt(d), 0.2 * rnd(d, seed=7)
    ref = _win_block_ref(x, g1, b1, wqkv, bqkv, wproj, bproj, nh)
    dev = [t.cuda() for t in (wqkv, bqkv, wproj, bproj)]
    img, bias = ops.window_block_pack(*dev, nh)
    xg = x.cuda()
    y = ops.window_block_fwd(xg, g1.cuda(), b1.cuda(), img, bias, nh)
    # the residual x is carried in fp32: compare the bf16-computed update, not x + update
    assert rel_err(y.cpu() - x, ref - x) < 1e-2
    y2 = ops.window_block_fwd(xg, g1.cuda(), b1.cuda(), img, bias, nh, out=xg)   # in place
    assert torch.equal(y2, y)


def test_fused_window_block_rejects_unsupported(ops):
    from image_segmentation_amd._C import LsamError
    d = 112
    w = [rnd(3 * d, d).cuda(), rnd(3 * d).cuda(), rnd(d, d).cuda(), rnd(d).cuda()]
    img, bias = ops.window_block_pack(*w, 2)
    x = rnd(1, 12, 12, d).cuda()   # 12 is not a multiple of the 8x8 window
    with pytest.raises(LsamError):
        ops.window_block_fwd(x, rnd(d).cuda(), rnd(d).cuda(), img, bias, 2)
    with pytest.raises(LsamError):
        ops.window_block_pack(rnd(12 * d, 4 * d).cuda(), rnd(12 * d).cuda(), rnd(4 * d, 4 * d).cuda(), rnd(4 * d).cuda(), 8)


@pytest.mark.parametrize("d,M", [(112, 1000), (56, 300), (224, 777), (64, 256), (160, 100), (144, 1000), (136, 130)])
def test_fused_mlp_block(ops, d, M):
    x = rnd(M, d, seed=11)
    g2, b2 = 1 + 0.1 * rnd(d, seed=12), 0.1 * rnd(d, seed=13)
    w1, c1 = rnd(4 * d, d, seed=14) / math.sqrt(d), 0.2 * rnd(4 * d, seed=15)
    w2, c2 = rnd(d, 4 * d, seed=16) / math.sqrt(4 * d), 0.2 * rnd(d, seed=17)
    ref = x + F.gelu(F.layer_norm(x, (d,), g2, b2, 1e-6) @ w1.t() + c1) @ w2.t() + c2
    img, bias = ops.mlp_block_pack(w1.cuda(), c1.cuda(), w2.cuda(), c2.cuda())
    xg = x.cuda()
    y = ops.mlp_block_fwd(xg, g2.cuda(), b2.cuda(), img, bias)
    assert rel_err(y.cpu() - x, ref - x) < 1e-2
    y2 = ops.mlp_block_fwd(xg, g2.cuda(), b2.cuda(), img, bias, out=xg)   # in place
    assert torch.equal(y2, y)


@pytest.mark.parametrize("d,M", [(112, 4096 + 17), (56, 300), (224, 777), (144, 2049)])
def test_fused_mlp_block_with_next_norm(ops, d, M):
    """Second output of the fused MLP block: LayerNorm of its result with the next block's norm1 (hieradet.py:136)."""
    x = rnd(M, d, seed=21)
    g2, b2 = 1 + 0.1 * rnd(d, seed=22), 0.1 * rnd(d, seed=23)
    gn, bn = (1 + 0.1 * rnd(d, seed=24)).cuda(), (0.1 * rnd(d, seed=25)).cuda()
    w1, c1 = rnd(4 * d, d, seed=26) / math.sqrt(d), 0.2 * rnd(4 * d, seed=27)
    w2, c2 = rnd(d, 4 * d, seed=28) / math.sqrt(4 * d), 0.2 * rnd(d, seed=29)
    img, bias = ops.mlp_block_pack(w1.cuda(), c1.cuda(), w2.cuda(), c2.cuda())
    y_plain = ops.mlp_block_fwd(x.cuda(), g2.cuda(), b2.cuda(), img, bias)
    y, xn = ops.mlp_block_fwd_next_norm(x.cuda(), g2.cuda(), b2.cuda(), img, bias, gn, bn)
    assert torch.equal(y, y_plain)   # the first output is unchanged, bit for bit
    want = F.layer_norm(y, (d,), gn, bn, 1e-6)
    assert rel_err(xn.float().cpu(), want.cpu()) < 1e-2   # bf16 rounding of the output only
    with pytest.raises(Exception):
        ops.mlp_block_fwd_next_norm(x.cuda(), g2.cuda(), b2.cuda(), img, bias, gn[1:], bn)   # misaligned gamma


@pytest.mark.parametrize("nh,hd,N,H,with_next", [(2, 56, 2, 32, True), (1, 56, 3, 16, False), (2, 32, 1, 24, True), (1, 32, 2, 8, False),
                                                 (2, 56, 5, 40, False)])
def test_stage1_block_split_projection(ops, nh, hd, N, H, with_next):
    """A whole stage-1 block (hieradet.py:134-166) as the plan runs it: the attention kernel stops at its output rows and
    proj + residual open the MLP kernel.  Against fp32 torch, and against the two kernels with the projection where it was
    (the same products in another order: close, not bit-equal)."""
    d = nh * hd
    x = rnd(N, H, H, d, seed=31)
    g1, b1 = 1 + 0.1 * rnd(d, seed=32), 0.1 * rnd(d, seed=33)
    wqkv, bqkv = rnd(3 * d, d, seed=34) / math.sqrt(d), 0.2 * rnd(3 * d, seed=35)
    wproj, bproj = rnd(d, d, seed=36) / math.sqrt(d), 0.2 * rnd(d, seed=37)
    g2, b2 = 1 + 0.1 * rnd(d, seed=38), 0.1 * rnd(d, seed=39)
    w1, c1 = rnd(4 * d, d, seed=40) / math.sqrt(d), 0.2 * rnd(4 * d, seed=41)
    w2, c2 = rnd(d, 4 * d, seed=42) / math.sqrt(4 * d), 0.2 * rnd(d, seed=43)
    gn, bn = 1 + 0.1 * rnd(d, seed=44), 0.1 * rnd(d, seed=45)
    y1_ref = _win_block_ref(x, g1, b1, wqkv, bqkv, wproj, bproj, nh)
    y_ref = y1_ref + F.gelu(F.layer_norm(y1_ref, (d,), g2, b2, 1e-6) @ w1.t() + c1) @ w2.t() + c2
    wimg, wbias = ops.window_block_pack(wqkv.cuda(), bqkv.cuda(), wproj.cuda(), bproj.cuda(), nh)
    mimg, mbias = ops.mlp_block_pack(w1.cuda(), c1.cuda(), w2.cuda(), c2.cuda())
    pimg, pbias = ops.mlp_block_proj_pack(wproj.cuda(), bproj.cuda())
    xg = x.cuda()
    a = ops.window_block_attention_fwd(xg, g1.cuda(), b1.cuda(), wimg, wbias, nh)
    assert a.dtype == torch.bfloat16 and a.shape == x.shape
    out = ops.mlp_block_fwd_proj(xg.view(-1, d), a.view(-1, d), g2.cuda(), b2.cuda(), mimg, mbias, pimg, pbias,
                                 next_norm=(gn.cuda(), bn.cuda()) if with_next else None)
    y = out[0] if with_next else out
    assert rel_err(y.cpu().view_as(x) - x, y_ref - x) < 1e-2
    # the un-split pair of kernels
    y1 = ops.window_block_fwd(xg, g1.cuda(), b1.cuda(), wimg, wbias, nh)
    y_old = ops.mlp_block_fwd(y1.view(-1, d), g2.cuda(), b2.cuda(), mimg, mbias)
    assert rel_err((y - xg.view(-1, d)).cpu(), (y_old - xg.view(-1, d)).cpu()) < 3e-3
    if with_next:
        want = F.layer_norm(y, (d,), gn.cuda(), bn.cuda(), 1e-6)
        assert rel_err(out[1].float().cpu(), want.cpu()) < 1e-2
    assert torch.equal(xg.cpu(), x)     # the residual stream is not written by either kernel here


def test_stage1_block_split_projection_rejects_unsupported(ops):
    from image_segmentation_amd._C import LsamError
    with pytest.raises(LsamError):
        ops.mlp_block_proj_pack(rnd(224, 224).cuda(), rnd(224).cuda())   # the W_proj fragments of a 224-wide block do not fit a chunk


def test_fused_mlp_block_rejects_unsupported(ops):
    from image_segmentation_amd._C import LsamError
    with pytest.raises(LsamError):
        ops.mlp_block_pack(rnd(1792, 448).cuda(), rnd(1792).cuda(), rnd(448, 1792).cuda(), rnd(448).cuda())


@pytest.mark.parametrize("d,M,with_next", [(448, 1000, True), (224, 777, True), (448, 128, False), (224, 4096, False)])
def test_fused_tail_block(ops, d, M, with_next):
    """y = res + proj(a); out = y + MLP(LN2(y)); optional LN1 of the next block (hieradet.py:79,163-165,136)."""
    res = rnd(M, d, seed=21)
    a = rnd(M, d, seed=22).bfloat16()
    wp, bp = rnd(d, d, seed=23) / math.sqrt(d), 0.2 * rnd(d, seed=24)
    g2, b2 = 1 + 0.1 * rnd(d, seed=25), 0.1 * rnd(d, seed=26)
    w1, c1 = rnd(4 * d, d, seed=27) / math.sqrt(d), 0.2 * rnd(4 * d, seed=28)
    w2, c2 = rnd(d, 4 * d, seed=29) / math.sqrt(4 * d), 0.2 * rnd(d, seed=30)
    gn, bn = 1 + 0.1 * rnd(d, seed=31), 0.1 * rnd(d, seed=32)
    y = res + a.float() @ wp.t() + bp
    ref = y + F.gelu(F.layer_norm(y, (d,), g2, b2, 1e-6) @ w1.t() + c1) @ w2.t() + c2
    img, vec = ops.tail_block_pack(*[t.cuda() for t in (wp, bp, g2, b2, w1, c1, w2, c2)])
    rg = res.cuda()
    got = ops.tail_block_fwd(a.cuda(), rg, img, vec, next_norm=(gn.cuda(), bn.cuda()) if with_next else None)
    out, xn = got if with_next else (got, None)
    # the residual is carried in fp32: compare the bf16-computed update
    assert rel_err(out.cpu() - res, ref - res) < 1e-2
    if with_next:
        assert rel_err(xn.float().cpu(), F.layer_norm(out.cpu(), (d,), gn, bn, 1e-6)) < 5e-3
    out2 = ops.tail_block_fwd(a.cuda(), rg, img, vec, out=rg)   # in place on the residual stream
    assert torch.equal(out2, out)


@pytest.mark.parametrize("e0,N,S", [(112, 2, 64), (112, 1, 256), (144, 2, 96), (112, 3, 40)])
def test_patch_embed_direct(ops, e0, N, S):
    """PatchEmbed conv 7x7/4 pad 3 + bias + pos embed, NHWC out (backbones/utils.py:85-93, hieradet.py:307-312):
    the direct-convolution kernel against F.conv2d on the same bf16-rounded operands (fp32 accumulation both
    sides, so only the summation order differs).  S = 40 / 64 / 96 leave partial 8 x 32 output tiles."""
    G = S // 4
    frames = rnd(N, 3, S, S, seed=41)
    w = rnd(e0, 3, 7, 7, seed=42) / 12.0
    b = 0.2 * rnd(e0, seed=43)
    pos = 0.5 * rnd(G, G, e0, seed=44)
    ref = F.conv2d(frames.bfloat16().float(), w.bfloat16().float(), b, stride=4, padding=3).permute(0, 2, 3, 1) + pos
    img = ops.patch_embed_pack(w.cuda())
    got = ops.patch_embed_fwd(frames.cuda(), img, b.cuda(), pos.cuda())
    assert got.shape == (N, G, G, e0)
    assert rel_err(got.cpu(), ref) < 1e-5


def test_patch_embed_direct_rejects_unsupported(ops):
    from image_segmentation_amd._C import LsamError
    with pytest.raises(LsamError):
        ops.patch_embed_pack(rnd(160, 3, 7, 7).cuda())


@pytest.mark.parametrize("n_img,H,Wd,N,K", [(2, 16, 32, 224, 112), (1, 8, 64, 448, 224), (2, 4, 128, 224, 112),
                                            (1, 256, 256, 224, 112), (3, 8, 32, 896, 448),
                                            # more tiles than workgroups: the persistent kernel's tile-to-tile path
                                            (3, 256, 128, 448, 224), (5, 128, 128, 224, 64),
                                            # Hiera-L widths: column-guarded 224-wide tiles (288) and 192-wide tiles (576, 1152)
                                            (2, 64, 64, 288, 144), (1, 32, 128, 576, 288), (2, 16, 32, 1152, 576),
                                            (1, 8, 32, 152, 40)])
def test_linear_maxpool2x2(ops, n_img, H, Wd, N, K):
    """shortcut = do_pool(proj(x_norm)) (hieradet.py:139-140, 27-41) with the pool in the GEMM epilogue, against
    F.max_pool2d of the un-pooled fp32 product of the same bf16 operands."""
    A = rnd(n_img * H * Wd, K, seed=51).bfloat16()
    W = (rnd(N, K, seed=52) / math.sqrt(K)).bfloat16()
    b = 0.3 * rnd(N, seed=53)
    full = (A.float() @ W.float().t() + b).view(n_img, H, Wd, N).permute(0, 3, 1, 2)
    ref = F.max_pool2d(full, 2, 2).permute(0, 2, 3, 1)
    got = ops.linear_maxpool2x2(A.cuda(), W.cuda(), b.cuda(), n_img, H, Wd)
    assert got.shape == ref.shape
    assert rel_err(got.cpu(), ref) < 1e-5


def test_linear_maxpool2x2_rejects_unsupported(ops):
    from image_segmentation_amd._C import LsamError
    A, W, b = rnd(16 * 16, 112).bfloat16().cuda(), rnd(224, 112).bfloat16().cuda(), rnd(224).cuda()
    with pytest.raises(LsamError):
        ops.linear_maxpool2x2(A, W, b, 1, 16, 16)   # 16-wide grid: no whole 32-column patch


@pytest.mark.parametrize("M,K", [(4096, 112), (1000, 224), (777, 144), (64, 16)])
def test_lateral_conv(ops, M, K):
    """FpnNeck lateral 1x1 conv (image_encoder.py:107-110) from the fp32 stage output: against the fp32 product of
    the bf16-rounded operands, rounded to bf16."""
    x = rnd(M, K, seed=61)
    w = rnd(256, K, seed=62) / math.sqrt(K)
    b = 0.3 * rnd(256, seed=63)
    ref = (x.bfloat16().float() @ w.bfloat16().float().t() + b)
    got = ops.lateral_conv(x.cuda(), w.cuda(), b.cuda())
    assert got.dtype == torch.bfloat16 and got.shape == (M, 256)
    assert rel_err(got.float().cpu(), ref) < 3e-3   # bf16 output rounding


def test_lateral_conv_unsupported_widths(ops):
    assert not ops.lateral_conv_supported(448, 256) and not ops.lateral_conv_supported(112, 128)
    assert ops.lateral_conv_supported(112, 256) and ops.lateral_conv_supported(224, 256)


def test_fused_tail_block_rejects_unsupported(ops):
    from image_segmentation_amd._C import LsamError
    d = 112
    with pytest.raises(LsamError):
        ops.tail_block_pack(rnd(d, d).cuda(), rnd(d).cuda(), rnd(d).cuda(), rnd(d).cuda(), rnd(4 * d, d).cuda(),
                            rnd(4 * d).cuda(), rnd(d, 4 * d).cuda(), rnd(d).cuda())


@pytest.mark.parametrize("N,H,W,dim,hd", [(2, 16, 24, 112, 56), (1, 8, 8, 96, 64), (3, 32, 16, 128, 24)])
def test_kv_window_attention_fused(N, H, W, dim, hd):
    """q-pool stage-entry block, 8x8 windows (hieradet.py:56-81 with do_pool(q) :69-71): k | v projected inside the
    attention kernel against a plain fp32 torch evaluation of the same block half from the same bf16 inputs."""
    import torch.nn.functional as F
    from image_segmentation_amd import ops
    nh, q = 4, 4 * hd
    g = torch.Generator().manual_seed(dim + hd)
    xn = torch.randn(N, H, W, dim, generator=g).bfloat16()
    w = (torch.randn(3 * q, dim, generator=g) / dim ** 0.5).bfloat16().float()
    b = 0.2 * torch.randn(3 * q, generator=g)
    x32 = xn.float()
    qkv = x32 @ w.t() + b                                                   # [N, H, W, 3q]
    qf, kf, vf = qkv[..., :q], qkv[..., q:2 * q].bfloat16().float(), qkv[..., 2 * q:].bfloat16().float()
    qp = F.max_pool2d(qf.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous()   # do_pool(q)
    def windows(t, ws):                                                      # [N, h, w, C] -> [N*nwin, ws*ws, nh, hd]
        n, hh, ww, c = t.shape
        t = t.view(n, hh // ws, ws, ww // ws, ws, nh, c // nh).permute(0, 1, 3, 2, 4, 5, 6)
        return t.reshape(-1, ws * ws, nh, c // nh)
    qw, kw, vw = windows(qp.bfloat16().float(), 4), windows(kf, 8), windows(vf, 8)
    o = F.scaled_dot_product_attention(qw.transpose(1, 2), kw.transpose(1, 2), vw.transpose(1, 2))   # [B, nh, 16, hd]
    o = o.transpose(1, 2).reshape(N, H // 8, W // 8, 4, 4, q).permute(0, 1, 3, 2, 4, 5).reshape(N, H // 2, W // 2, q)
    img = ops.kv_window_attention_pack(w.cuda(), b.cuda(), nh)
    got = ops.kv_window_attention_fwd(xn.cuda(), qp.cuda(), img, nh)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (N, H // 2, W // 2, q)
    assert rel_err(got.float(), o) < 1e-2
    # unsupported geometries are refused, not rerouted
    from image_segmentation_amd._C import LsamError
    with pytest.raises(LsamError):
        ops.kv_window_attention_fwd(xn.cuda()[:, :, :W - 4].contiguous(), qp.cuda()[:, :, :W // 2 - 2].contiguous(), img, nh)


@pytest.mark.parametrize("N,H,W,dim,nh,hd,pooled", [(2, 16, 24, 224, 4, 56, False), (1, 4, 4, 224, 4, 56, False),
                                                    (3, 20, 12, 96, 2, 48, False), (1, 36, 28, 64, 1, 64, False),
                                                    (2, 16, 24, 224, 8, 56, True), (1, 4, 4, 112, 4, 56, True),
                                                    (3, 20, 12, 200, 3, 24, True)])
def test_qkv_window4_attention_fused(N, H, W, dim, nh, hd, pooled):
    """4x4-window blocks (hieradet.py:56-81; with do_pool(q) :69-71 for the stage-entry variant): qkv (or k | v) projected
    inside the attention kernel, against a plain fp32 torch evaluation of the same block half from the same bf16 inputs.
    Window counts that are not a multiple of the windows a wave takes per pass, one / two heads per workgroup."""
    import torch.nn.functional as F
    from image_segmentation_amd import ops
    q = nh * hd
    g = torch.Generator().manual_seed(dim + hd + H)
    xn = torch.randn(N, H, W, dim, generator=g).bfloat16()
    w = (torch.randn(3 * q, dim, generator=g) / dim ** 0.5).bfloat16().float()
    b = 0.2 * torch.randn(3 * q, generator=g)
    qkv = xn.float() @ w.t() + b                                             # [N, H, W, 3q]
    qf, kf, vf = qkv[..., :q], qkv[..., q:2 * q].bfloat16().float(), qkv[..., 2 * q:].bfloat16().float()
    def windows(t, ws):                                                      # [N, h, w, C] -> [N*nwin, ws*ws, nh, hd]
        n, hh, ww, c = t.shape
        t = t.view(n, hh // ws, ws, ww // ws, ws, nh, c // nh).permute(0, 1, 3, 2, 4, 5, 6)
        return t.reshape(-1, ws * ws, nh, c // nh)
    kw, vw = windows(kf, 4), windows(vf, 4)
    img = ops.qkv_window4_attention_pack(w.cuda(), b.cuda(), nh, pooled_q=pooled)
    if pooled:
        qp = F.max_pool2d(qf.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous()   # do_pool(q)
        qw = windows(qp.bfloat16().float(), 2)
        o = F.scaled_dot_product_attention(qw.transpose(1, 2), kw.transpose(1, 2), vw.transpose(1, 2))   # [B, nh, 4, hd]
        o = o.transpose(1, 2).reshape(N, H // 4, W // 4, 2, 2, q).permute(0, 1, 3, 2, 4, 5).reshape(N, H // 2, W // 2, q)
        got = ops.qkv_window4_attention_fwd(xn.cuda(), img, nh, hd, q_pooled=qp.cuda())
    else:
        qw = windows(qf.bfloat16().float(), 4)
        o = F.scaled_dot_product_attention(qw.transpose(1, 2), kw.transpose(1, 2), vw.transpose(1, 2))   # [B, nh, 16, hd]
        o = o.transpose(1, 2).reshape(N, H // 4, W // 4, 4, 4, q).permute(0, 1, 3, 2, 4, 5).reshape(N, H, W, q)
        got = ops.qkv_window4_attention_fwd(xn.cuda(), img, nh, hd)
    assert got.dtype == torch.bfloat16 and got.shape == o.shape
    assert rel_err(got.float(), o) < 1e-2
    # unsupported geometries are refused, not rerouted
    from image_segmentation_amd._C import LsamError
    with pytest.raises(LsamError):
        ops.qkv_window4_attention_fwd(xn.cuda()[:, :, :W - 2].contiguous(), img, nh, hd)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K", [(16384, 896), (2000, 2688), (1030, 120), (1024, 40)])
def test_linear_skinny32(dt, M, K):
    """N = 32 products (the LoRA u / t columns) take the streaming kernel; written in place behind the operand's K columns,
    as the plan does (xn[:, d:d+32] = xn[:, :d] . A^T)."""
    from image_segmentation_amd import ops
    g = torch.Generator().manual_seed(M + K)
    buf = torch.randn(M, K + 32, generator=g).to(dt).cuda()
    w = (torch.randn(32, K, generator=g) / K ** 0.5).to(dt).cuda()
    a = buf[:, :K]
    ref = a.double() @ w.double().t()
    out = ops.linear(a, w, out=buf[:, K:])
    assert out.data_ptr() == buf[:, K:].data_ptr()
    assert rel_err(buf[:, K:].double(), ref) < (1e-6 if dt == torch.float32 else 5e-3)
    assert torch.equal(buf[:, :K], a)                                   # the operand columns are untouched
    o32 = ops.linear(a, w, out_f32=True)
    assert o32.dtype == torch.float32 and rel_err(o32.double(), ref) < (1e-6 if dt == torch.float32 else 1e-5 + 0)


@pytest.mark.parametrize("M,K,N", [(262144 // 8, 224, 672), (70001, 224, 896), (5000, 112, 448), (64, 144, 432 + 16), (1, 16, 32),
                                   (33000, 200, 1344), (4097, 224, 64)])
def test_stream_linear(M, K, N):
    """The small-K Linear with LDS-resident weights (stage-2 qkv, hieradet.py:59): one / several / partial column groups,
    ragged row counts, operands and results inside wider buffers (the plan writes k | v behind the q columns), and the
    same bits as lsam_linear (bias first, k-steps in order)."""
    from image_segmentation_amd import ops
    g = torch.Generator().manual_seed(M + K + N)
    abuf = torch.randn(M, K + 24, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    obuf = torch.full((M, N + 40), 7.0, dtype=torch.bfloat16, device="cuda")
    a = abuf[:, :K]
    img = ops.stream_linear_pack(w)
    out = ops.stream_linear_fwd(a, img, b, K, N, out=obuf[:, 8:8 + N])
    ref = a.double() @ w.bfloat16().double().t() + b.double()
    assert rel_err(out.double(), ref) < 5e-3
    assert (obuf[:, :8] == 7.0).all() and (obuf[:, 8 + N:] == 7.0).all()          # nothing outside the N columns is written
    tiled = ops.linear(a, ops.pack_matrix(w, torch.bfloat16), b, K=K)
    if M >= 1024 and N % 224 == 0:      # the persistent tiled kernels seed their accumulators with the bias too: the same sums in the same order
        assert torch.equal(out, tiled)
    else:              # (the generic small-M kernel adds the bias last)
        assert rel_err(out.double(), tiled.double()) < 4e-3


@pytest.mark.parametrize("M,N,K,few", [(1040, 32, 44, False), (2048, 32, 900, False), (37, 48, 12, True), (130, 256, 260, True),
                                       (16, 8, 4, True)])
def test_linear_streaming_fp32_k_not_multiple_of_8(M, N, K, few):
    """fp32 operands need K % 4 == 0 only; an fp32 fragment is two 16-byte chunks, and for K % 8 == 4 the second chunk of the
    last fragment lies past K.  The operand rows here are wider than K and hold NaN there (and the row after the last is NaN
    too), so a kernel that reads past K multiplies NaNs into the result.  Both routes into the streaming kernel: N = 32
    (the LoRA u / t products) and LSAM_EPI_FEW_ROWS."""
    from image_segmentation_amd import ops
    assert K % 8 == 4
    g = torch.Generator().manual_seed(M * 7 + K)
    abuf = torch.full((M + 1, K + 12), float("nan"))
    wbuf = torch.full((N + 1, K + 12), float("nan"))
    abuf[:M, :K] = torch.randn(M, K, generator=g)
    wbuf[:N, :K] = torch.randn(N, K, generator=g) / K ** 0.5
    abuf, wbuf = abuf.cuda(), wbuf.cuda()
    a, w = abuf[:M], wbuf[:N]
    ref = a[:, :K].double() @ w[:, :K].double().t()
    out = ops.linear(a, w, K=K, few_rows=few)
    assert torch.isfinite(out).all()
    assert rel_err(out.double(), ref) < 2e-6


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,epi", [(8, 256, 256, "relu"), (130, 2048, 256, "none"), (37, 8, 256, "sigmoid"), (128, 256, 2048, "res"),
                                       (5, 40, 8, "none"), (4100, 128, 264, "res")])
def test_linear_few_rows(dt, M, N, K, epi):
    """LSAM_EPI_FEW_ROWS: the streaming kernel with the bias / activation / residual epilogue, ragged M and N, and the same
    rows whatever the batch around them (the K split never depends on M)."""
    from image_segmentation_amd import ops
    g = torch.Generator().manual_seed(M * N + K)
    a = torch.randn(M, K, generator=g).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda() if epi == "res" else None
    ref = a.double() @ w.double().t() + b.double()
    ref = ref.relu() if epi == "relu" else ref.sigmoid() if epi == "sigmoid" else ref + res.double() if epi == "res" else ref
    out = ops.linear(a, w, b, res=res, relu=epi == "relu", sigmoid=epi == "sigmoid", out_f32=True, few_rows=True)
    assert rel_err(out.double(), ref) < (2e-6 if dt == torch.float32 else 1e-5)
    tile = ops.linear(a, w, b, res=res, relu=epi == "relu", sigmoid=epi == "sigmoid", out_f32=True)
    assert rel_err(out, tile) < 2e-6
    k = min(M, 3)
    part = ops.linear(a[:k], w, b, res=None if res is None else res[:k], relu=epi == "relu", sigmoid=epi == "sigmoid", out_f32=True, few_rows=True)
    assert torch.equal(part, out[:k])
    if dt == torch.bfloat16:
        ot = ops.linear(a, w, b, res=res, relu=epi == "relu", sigmoid=epi == "sigmoid", few_rows=True)
        assert ot.dtype == dt and rel_err(ot.double(), ref) < 5e-3
    with pytest.raises(Exception):
        ops.linear(a, w, b, gelu=True, few_rows=True)
