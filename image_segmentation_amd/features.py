"""Encoder-only feature-extraction pass (BASELINE.json configs[4]): every frame goes through
`Hiera.get_embedding` (hieradet.py:284-302: all blocks under no_grad, LoRA factors applied) and is
reduced to the spatial mean of its last-stage feature map (896-d for Hiera-B+).  These per-frame /
per-video embeddings are what the reference's nearest-domain LoRA selection consumes
(`set_closest_lora`, train_cl_kmean.py:189-207: L2 distance to per-domain anchors, argmin).

Frames are independent, so the pass shards by frame with NO data-path collective: rank r owns a
contiguous slice of the frame index range and writes only its rows; `gather_features` is the one
optional exchange at the very end (all_gather of [n_local, C] fp32 rows)."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import ops


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of range(n_items) for `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world) or n_items < 0:
        raise ValueError(f"shard_range: rank {rank} of {world}, {n_items} items")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@torch.no_grad()
def extract_batch(trunk, frames: torch.Tensor, pool: Optional[Callable] = None) -> torch.Tensor:
    """One batch of the pass: frames [n,3,S,S] -> fp32 [n, C] = spatial mean of Hiera.get_embedding's last stage."""
    pool = ops.spatial_mean if pool is None else pool
    last = trunk.get_embedding(frames)[-1]                              # NCHW view of NHWC storage
    return pool(last.permute(0, 2, 3, 1).contiguous())                  # (already contiguous: no copy)


@torch.no_grad()
def extract_features(trunk, frame_source: Callable[[int, int], torch.Tensor], n_frames: int, batch: int,
                     rank: int = 0, world: int = 1, pool: Optional[Callable] = None) -> Tuple[torch.Tensor, int]:
    """Pooled last-stage embedding of frames [lo, hi) = shard_range(n_frames, rank, world).

    frame_source(first, n) returns frames [n,3,S,S] (already on the trunk's device).  Returns
    (features fp32 [hi-lo, C], lo).  `pool` defaults to the HIP kernel ops.spatial_mean."""
    if batch <= 0:
        raise ValueError("extract_features: batch must be positive")
    lo, hi = shard_range(n_frames, rank, world)
    rows = []
    for first in range(lo, hi, batch):
        n = min(batch, hi - first)
        rows.append(extract_batch(trunk, frame_source(first, n), pool))
    if not rows:
        c = trunk.channel_list[0]
        dev = next(trunk.parameters()).device
        return torch.empty(0, c, dtype=torch.float32, device=dev), lo
    return torch.cat(rows, 0), lo


def gather_features(local: torch.Tensor, n_frames: int, group=None) -> torch.Tensor:
    """All ranks' rows in frame order, on every rank: [n_frames, C].  Single-process: identity."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(n_frames, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(width, local.shape[1], dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], 0)


def closest_domain(video_embedding: torch.Tensor, anchors: Dict[str, torch.Tensor],
                   available: Optional[Sequence[str]] = None) -> Tuple[str, Dict[str, float]]:
    """train_cl_kmean.py:189-207: the domain whose anchor is nearest (L2) to the video embedding,
    first minimum on ties, among `available` (default: all anchors in insertion order)."""
    names = list(anchors) if available is None else list(available)
    if not names:
        raise ValueError("closest_domain: no anchors")
    v = video_embedding.detach().float().cpu()
    dists = {n: float((anchors[n].detach().float().cpu() - v).norm()) for n in names}
    best = names[0]
    for n in names[1:]:
        if dists[n] < dists[best]:
            best = n
    return best, dists
