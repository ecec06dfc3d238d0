#!/usr/bin/env python3
"""Throughput of the LoRA-SAM2 Hiera-B+ image-encoder fwd+bwd hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch of synthetic frames already resident in
HBM: encoder forward (trunk + FPN neck) under bf16, loss = sum_l mean(fpn_l^2), backward to
the LoRA A/B factors, mean all-reduce of the LoRA gradients (N>1), AdamW step on them.
Workload = BASELINE.json configs[1]: sam2.1_hiera_base_plus + LoRA_qkv r=4, bf16, batch 16 per
GPU (weak scaling: configs[2] is 16 x 8 = 128).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
import types

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GF_PER_FRAME = 657.4e9       # algorithmic fwd+bwd FLOPs, Hiera-B+ r=4 @1024^2
PEAK_BF16_TFLOPS = 2500.0    # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
BPLUS = dict(embed_dim=112, num_heads=2)
# dominant kernel of the step (rocprofv3 --kernel-trace --stats, profiles/): the fused block tail of the
# sixteen 448-wide stage-3 blocks (tailblock.hip).  liblsam tags its live event records with 9000 + d.
DOMINANT_CFG = 9448
DOMINANT_KERNEL = ("lsam::tailpair_kernel<14, 28> (proj + residual + LN2 + fc1 + GELU + fc2 + residual + "
                   "next LN1, 448 wide; algorithmic FLOPs 18 M d^2 per launch)")


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/<round>/traffic.json, written by tools/pmc_traffic.py from separate FETCH_SIZE and
    WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py
    cannot run the counter passes on itself; null when the file is absent."""
    best = None
    root = os.path.join(REPO, "profiles")
    if os.path.isdir(root):
        for r in sorted(os.listdir(root)):
            f = os.path.join(root, r, "traffic.json")
            if os.path.isfile(f):
                best = f
    if best is None:
        return None
    try:
        with open(best) as fh:
            return json.load(fh).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def build(device, rank_seed=0, lora_rank=4):
    import image_segmentation_amd as m
    from image_segmentation_amd.synth import fill_synthetic_, synth_tensor
    enc = m.build_image_encoder(dict(BPLUS))
    fill_synthetic_(enc, 0)
    m.wrap_image_encoder_lora(types.SimpleNamespace(image_encoder=enc), lora_rank)
    with torch.no_grad():
        for name, p in enc.named_parameters():
            if "linear_a_" in name or "linear_b_" in name:
                p.copy_(synth_tensor(name, tuple(p.shape), 0))
    enc = enc.to(device)
    enc.trunk.compute_dtype = torch.bfloat16
    return enc


def fpn_loss(out):
    """sum_l mean(fpn_l^2), evaluated as ||f||_2^2 / numel with fp32 accumulation: one read of
    each level instead of materialising fp32 copies and squares of the 256x256x256 level."""
    return sum(torch.linalg.vector_norm(f, dtype=torch.float32) ** 2 / f.numel() for f in out["backbone_fpn"])


def cpu_baseline(iters: int):
    """The oracle (CPU restatement of the reference path, pinned to reference goldens) timed on
    this box's host cores: configs[0] = B+ r=4, batch 1, fp32, one 1024^2 frame, fwd+bwd."""
    from image_segmentation_amd.synth import synth_frames
    from oracle import hiera_oracle as O
    from oracle.params import make_params
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    spec = O.block_specs(**BPLUS)
    p = make_params(spec, 4, 0)
    img = synth_frames(1, 1024, 0)

    def it():
        for t in p.values():
            t.grad = None
        loss = O.fpn_loss(O.image_encoder_forward(p, img, spec))
        loss.backward()

    it()
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    dt = time.perf_counter() - t0
    return {"value": round(iters / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{iters} timed fwd+bwd iterations (1 warm-up) of configs[0]: B+ r=4, batch 1, "
                      f"fp32, one 1024x1024 frame, torch CPU kernels on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU")
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=device)

    from image_segmentation_amd import _C
    from image_segmentation_amd.dp import LoRAGradReducer
    from image_segmentation_amd.synth import synth_frames_device

    enc = build(device)
    lora = [p for p in enc.parameters() if p.requires_grad]
    reducer = LoRAGradReducer(lora).attach(enc.trunk)   # per-block buckets, all-reduce overlapped with backward
    reducer.broadcast_params(0)
    opt = torch.optim.AdamW(lora, lr=1e-4)   # config.yaml:6 of the reference
    B = args.batch
    img = synth_frames_device(B, 1024, 0, device, first_index=rank * B)   # resident before timing

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = enc(img)
        loss = fpn_loss(out)
        loss.backward()
        reducer.reduce()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(loss).item(), "non-finite loss"
    fps = world * B * args.steps / elapsed

    roofline = None
    if rank == 0:
        # live per-launch timing of the dominant kernel (gemm3_kernel<false, false>: the persistent
        # 256x224-tile bf16 MFMA GEMM with bf16 output that runs most block linears): HIP events on
        # the launch stream, recorded inside liblsam (cfg 2224), over `psteps` further steps of the
        # same workload.  achieved = algorithmic 2*M*N*K FLOPs of those launches / their duration.
        psteps = min(3, args.steps)
        cap = 400 * psteps
        _C.check(_C.lib().lsam_prof_arm(cap))
        for _ in range(psteps):
            step()
        torch.cuda.synchronize()
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        cfg = (C.c_int * cap)()
        n = _C.lib().lsam_prof_read(cap, ms, fl, cfg, None)
        sel = [i for i in range(n) if cfg[i] == DOMINANT_CFG]
        all_ms = sum(ms[i] for i in range(n))
        tot_ms = sum(ms[i] for i in sel)
        tot_fl = sum(fl[i] for i in sel)
        achieved = tot_fl / (tot_ms * 1e-3) / 1e12
        gemm = [i for i in range(n) if cfg[i] == 2224]
        gemm_tf = sum(fl[i] for i in gemm) / max(1e-9, sum(ms[i] for i in gemm) * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": DOMINANT_KERNEL,
                    "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": measured_traffic(),
                    "launches_per_step": len(sel) // psteps,
                    "avg_launch_us": round(1e3 * tot_ms / max(1, len(sel)), 2),
                    "avg_launch_gflop": round(tot_fl / max(1, len(sel)) / 1e9, 3),
                    "algorithmic_hbm_bytes_per_launch": 12 * B * 64 * 64 * 448,
                    "kernel_ms_per_step": round(tot_ms / psteps, 3),
                    "second_kernel": {"kernel": "lsam::gemm3_kernel<false, false> (256x224 persistent bf16 GEMM)",
                                      "achieved": round(gemm_tf, 2), "launches_per_step": len(gemm) // psteps,
                                      "kernel_ms_per_step": round(sum(ms[i] for i in gemm) / psteps, 3)},
                    "all_timed_mfma_ms_per_step": round(all_ms / psteps, 3)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        line = {
            "metric": "frames/sec LoRA-SAM2 Hiera-B+ fwd+bwd @1024^2",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic N(0,1) frames generated on-device; seeded synthetic weights (no checkpoints offline)",
            "config": {"workload": "BASELINE.json configs[1]: sam2.1_hiera_base_plus + LoRA_qkv r=4, bf16 "
                                   "fwd+bwd (+LoRA-grad all-reduce, AdamW), 1024x1024",
                       "frames_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "residual_stream": "fp32"},
            "mfma_frac_whole_step": round(fps / world * GF_PER_FRAME / (PEAK_BF16_TFLOPS * 1e12), 4),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_iters)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
