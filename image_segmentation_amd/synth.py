"""Deterministic synthetic weights / frames.

There is no network on the build or GPU boxes, so Meta's SAM 2.1 checkpoints
(README.md:12 of the reference) are never available.  Every test, golden
fixture and benchmark therefore uses weights drawn from this generator.  Each
tensor gets its own CPU ``torch.Generator`` seeded from ``crc32(name) ^ seed``,
so the values depend only on (name, shape, seed) -- not on module iteration
order -- and the GPU box regenerates them bit-exactly.

Scales are chosen so activations stay O(1) through 24 blocks (fan-in scaled
matrices), attention logits are O(1) (so softmax is neither uniform nor
one-hot), biases are large enough that Hiera's zero-pad tokens (whose qkv equals
the bias) matter, and LoRA B is non-zero so that dA is non-zero
(the reference's zero-B init, lora_qkv.py:191-192, makes dA vanish at step 0).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode("utf-8")) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """fp32 CPU tensor for parameter ``name`` (state_dict-style dotted name)."""
    g = _gen(name, seed)
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    is_norm = ".norm1." in name or ".norm2." in name or name.startswith("norm")
    if "linear_a_" in name:  # LoRA A: kaiming_uniform_(a=sqrt(5)) -> U(-1/sqrt(fan_in), +)
        bound = 1.0 / math.sqrt(shape[1])
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    if "linear_b_" in name:  # LoRA B: non-zero on purpose
        return torch.randn(shape, generator=g) * 0.1
    if "pos_embed" in name:
        return torch.randn(shape, generator=g) * 0.25
    if is_norm and leaf == "weight":
        return 1.0 + torch.randn(shape, generator=g) * 0.1
    if is_norm and leaf == "bias":
        return torch.randn(shape, generator=g) * 0.1
    if leaf == "bias":
        return torch.randn(shape, generator=g) * 0.2
    if leaf == "weight" and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return torch.randn(shape, generator=g) * 0.02


@torch.no_grad()
def fill_synthetic_(module: torch.nn.Module, seed: int = 0, prefix: str = "") -> torch.nn.Module:
    """Overwrite every parameter of ``module`` in place with synthetic values.

    ``prefix`` is stripped from nothing and added to nothing: names are the
    module's own ``named_parameters`` names, so the same call on the reference's
    modules and on this package's drop-in modules (identical state_dict keys)
    produces identical weights."""
    for name, p in module.named_parameters():
        p.copy_(synth_tensor(prefix + name, tuple(p.shape), seed).to(p.dtype))
    return module


def synth_state_dict(names_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0
                     ) -> Dict[str, torch.Tensor]:
    return {n: synth_tensor(n, s, seed) for n, s in names_shapes}


def synth_frames(n: int, size: int = 1024, seed: int = 0, device="cpu",
                 dtype=torch.float32) -> torch.Tensor:
    """Synthetic ImageNet-normalised frames ~ N(0,1), [n,3,size,size].

    Generated per frame so frame i is the same tensor whatever the batch size or
    rank (data-parallel shards draw disjoint frame indices)."""
    out = torch.empty(n, 3, size, size, dtype=torch.float32)
    for i in range(n):
        out[i] = torch.randn(3, size, size, generator=_gen(f"frame.{i}", seed))
    return out.to(device=device, dtype=dtype)


@torch.no_grad()
def synth_frames_device(n: int, size: int, seed: int, device, first_index: int = 0) -> torch.Tensor:
    """On-device N(0,1) frames for throughput runs (values differ from
    ``synth_frames``; parity tests use the CPU generator above)."""
    g = torch.Generator(device=device)
    g.manual_seed(1234567 + seed * 7919 + first_index)
    return torch.randn(n, 3, size, size, generator=g, device=device, dtype=torch.float32)
