"""Drop-in mirror of the reference's lora_qkv.py (encoder side).

`LoRA_qkv` keeps the reference's constructor, attribute names and state_dict keys
(lora_qkv.py:84-100) so `wrap_image_encoder_lora` (ours or the reference's own, which only
needs the class it instantiates to be this one) and saved LoRA files interoperate.  Inside a
`Hiera` the module is a parameter container: the trunk call folds the rank-r update into the
frozen-qkv MFMA loop.  Called on its own it runs the same fused GEMM through `lsam_linear`
with autograd for x and the four LoRA factors.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from ._C import LsamError

AUG = 32  # [q part 16 | v part 16] extra K columns, as in csrc/hiera.hip


def _dt():
    from .hieradet import _compute_dtype
    return _compute_dtype(None)


class _LoRAQKVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, w, b, aq, bq, av, bv):
        dt = _dt()
        M, d = x2.shape
        n = w.shape[0]
        r = aq.shape[0]
        if r > 16:
            raise LsamError("LoRA rank > 16 is not supported")
        dev = x2.device
        # packed operands: xa = [x | u], wa = [W | B_q rows 0..d | B_v rows n-d..n]  (slice rule lora_qkv.py:106-107)
        xa = torch.zeros(M, d + AUG, dtype=dt, device=dev)
        xa[:, :d] = ops.cast(x2, dt)
        acat = torch.zeros(AUG, d, dtype=torch.float32, device=dev)
        acat[:r] = aq
        acat[16:16 + r] = av
        baug = torch.zeros(n, AUG, dtype=torch.float32, device=dev)
        baug[:d, :r] = bq
        baug[n - d:, 16:16 + r] = bv
        wa = ops.pack_matrix(torch.cat([w, baug], dim=1), dt)
        ops.linear(xa, ops.pack_matrix(acat, dt), K=d, out=xa[:, d:])
        out = ops.linear(xa, wa, b)
        ctx.save_for_backward(xa, w, aq, bq, av, bv)
        ctx.dt = dt
        return out

    @staticmethod
    def backward(ctx, g):
        xa, w, aq, bq, av, bv = ctx.saved_tensors
        dt = ctx.dt
        M, n = g.shape
        d = w.shape[1]
        r = aq.shape[0]
        dev = g.device
        ga = torch.zeros(M, n + AUG, dtype=dt, device=dev)
        ga[:, :n] = ops.cast(g.float().contiguous(), dt)
        bT = torch.zeros(AUG, n, dtype=torch.float32, device=dev)
        bT[:r, :d] = bq.t()
        bT[16:16 + r, n - d:] = bv.t()
        ops.linear(ga, ops.pack_matrix(bT, dt), K=n, out=ga[:, n:])          # t = [dq.B_q | dv.B_v]
        d_aq = ops.outer_reduce(xa[:, :d], ga[:, n:n + 16], r, transpose_out=True)
        d_av = ops.outer_reduce(xa[:, :d], ga[:, n + 16:], r, transpose_out=True)
        d_bq = ops.outer_reduce(ga[:, :d], xa[:, d:d + 16], r, transpose_out=False)
        d_bv = ops.outer_reduce(ga[:, n - d:n], xa[:, d + 16:], r, transpose_out=False)
        dx = None
        if ctx.needs_input_grad[0]:
            a_rows = torch.zeros(AUG, d, dtype=torch.float32, device=dev)
            a_rows[:r] = aq
            a_rows[16:16 + r] = av
            wT = ops.pack_matrix(torch.cat([w, a_rows], dim=0).t(), dt)       # [d, n + AUG]
            dx = ops.linear(ga, wT, out_f32=True)
        return dx, None, None, d_aq, d_bq, d_av, d_bv


class LoRA_qkv(nn.Module):
    """LoRA adaption of the fused qkv Linear, queries and values only (lora_qkv.py:69-109).

    qkv = W x + b;  qkv[..., :d] += B_q(A_q x);  qkv[..., -d:] += B_v(A_v x),  d = in_features.
    No bias, no alpha/r scaling on the LoRA branch."""

    def __init__(self, qkv, linear_a_q: nn.Module, linear_b_q: nn.Module, linear_a_v: nn.Module,
                 linear_b_v: nn.Module):
        super().__init__()
        self.qkv = qkv
        self.linear_a_q = linear_a_q
        self.linear_b_q = linear_b_q
        self.linear_a_v = linear_a_v
        self.linear_b_v = linear_b_v
        self.d_model = qkv.in_features
        # the reference also creates an unused `w_identity = eye(d)` attribute (lora_qkv.py:100);
        # it is not a buffer, never saved and never read, so it is not reproduced.

    def _forward_packed(self, xn: torch.Tensor) -> torch.Tensor:
        """xn: T [M, d] already in compute dtype -> qkv T [M, 3*dim_out] (no autograd)."""
        with torch.no_grad():
            return _LoRAQKVFn.apply(xn.float(), self.qkv.weight, self.qkv.bias, self.linear_a_q.weight,
                                    self.linear_b_q.weight, self.linear_a_v.weight, self.linear_b_v.weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.qkv.weight.requires_grad or self.qkv.bias.requires_grad:
            if torch.is_grad_enabled():
                raise LsamError("LoRA_qkv: the base qkv Linear must be frozen; liblsam has no weight-gradient "
                                "GEMM for frozen tensors by design")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        out = _LoRAQKVFn.apply(x2, self.qkv.weight, self.qkv.bias, self.linear_a_q.weight, self.linear_b_q.weight,
                               self.linear_a_v.weight, self.linear_b_v.weight)
        return out.view(*lead, out.shape[-1])


def wrap_image_encoder_lora(sam_model, rank: int):
    """lora_qkv.py:148-192: freeze the image encoder, wrap trunk.blocks[-2:] with LoRA_qkv,
    A ~ kaiming_uniform(a=sqrt(5)), B = 0."""
    assert rank > 0
    for param in sam_model.image_encoder.parameters():
        param.requires_grad = False
    for blk in sam_model.image_encoder.trunk.blocks[-2:]:
        base = blk.attn.qkv
        d_model = base.in_features
        mods = [nn.Linear(d_model, rank, bias=False), nn.Linear(rank, d_model, bias=False),
                nn.Linear(d_model, rank, bias=False), nn.Linear(rank, d_model, bias=False)]
        for m in mods:
            m.to(device=base.weight.device)
        nn.init.kaiming_uniform_(mods[0].weight, a=math.sqrt(5))
        nn.init.kaiming_uniform_(mods[2].weight, a=math.sqrt(5))
        nn.init.zeros_(mods[1].weight)
        nn.init.zeros_(mods[3].weight)
        blk.attn.qkv = LoRA_qkv(base, *mods)


def custom_save_lora_parameters(model, save_path, include_v: bool = False):
    """lora_qkv.py:212-219.  The reference's name filter matches only the q factors
    ('_a_q', 'A_q', '_b_q', 'B_q'), so its files never hold the v factors; that format is kept
    by default for interchange.  include_v=True also stores '_a_v'/'_b_v'."""
    keys = ["_a_q", "A_q", "_b_q", "B_q"] + (["_a_v", "A_v", "_b_v", "B_v"] if include_v else [])
    sd = {k: v for k, v in model.state_dict().items() if any(s in k for s in keys)}
    torch.save(sd, save_path)


def custom_load_lora_parameters(model, load_path):
    """lora_qkv.py:224-228 (strict=False load + missing-key check on the q factors)."""
    sd = torch.load(load_path, map_location=next(model.parameters()).device, weights_only=True)
    err = model.load_state_dict(sd, strict=False)
    for name in err.missing_keys:
        if any(s in name for s in ("_a_q", "A_q", "_b_q", "B_q")):
            raise ValueError(f"Custom LoRA Missing key in state_dict: {name}")


class LoRABank:
    """In-memory sets of LoRA factors, switched without touching the file system.

    The reference's continual-learning loop runs the previous domain's LoRA as a distillation teacher by
    saving the current factors to disk, loading the previous ones, running the model, and loading the current
    ones back -- per batch, with five barriers around the files (train_cl.py:209-228), and its nearest-domain
    selection loads one LoRA file per candidate domain (train_cl_kmean.py:189-207).  Here every set stays
    resident on the device; `activate` is eight small device-to-device copies into the module's LoRA
    parameters, and the next forward re-packs only the LoRA-dependent slices of the packed weights
    (lsam_hiera_pack_lora), so a teacher pass costs one forward and no I/O.  All factors are kept, q and v
    (the reference's files hold only the q factors, see custom_save_lora_parameters)."""

    _KEYS = ("linear_a_q", "linear_b_q", "linear_a_v", "linear_b_v", "A_q", "B_q", "A_v", "B_v")

    def __init__(self, model: nn.Module):
        self._params = {n: p for n, p in model.named_parameters() if any(k in n for k in self._KEYS)}
        if not self._params:
            raise ValueError("LoRABank: the model has no LoRA parameters (wrap_image_encoder_lora first)")
        self._sets = {}

    def keys(self):
        return list(self._sets)

    def __contains__(self, key) -> bool:
        return key in self._sets

    @torch.no_grad()
    def store(self, key) -> None:
        """Snapshot the model's current LoRA factors under `key` (overwrites an existing set in place)."""
        old = self._sets.get(key)
        if old is not None:
            for n, p in self._params.items():
                old[n].copy_(p)
        else:
            self._sets[key] = {n: p.detach().clone() for n, p in self._params.items()}

    @torch.no_grad()
    def activate(self, key) -> None:
        """Make set `key` the model's LoRA factors (the parameters keep their identity, optimizer state and
        requires_grad; only their values change)."""
        if key not in self._sets:
            raise KeyError(f"LoRABank: no set {key!r} (have {self.keys()})")
        for n, p in self._params.items():
            p.copy_(self._sets[key][n])

    def state_dict(self, key):
        """The stored set as a state_dict fragment (load with model.load_state_dict(..., strict=False))."""
        return {n: t.clone() for n, t in self._sets[key].items()}

    def load_state_dict(self, key, sd) -> None:
        """Add a set from a (possibly q-only, reference-format) LoRA state_dict; missing v factors keep the
        model's current values, as custom_load_lora_parameters would leave them."""
        cur = {n: p.detach().clone() for n, p in self._params.items()}
        for n, t in sd.items():
            if n in cur:
                cur[n].copy_(t.to(cur[n].device, cur[n].dtype))
        self._sets[key] = cur
