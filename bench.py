#!/usr/bin/env python3
"""Throughput of the LoRA-SAM2 Hiera image-encoder hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config 1|3|4]
  N>1 either way: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
  bench.py --gpus N ...` (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or plain
  `python bench.py --gpus N`, which starts exactly that launcher as a CHILD process before anything touches the GPU and
  relays rank 0's line.  Under the launcher a process group is built for ANY N, so `--nproc-per-node 1` runs the whole
  RCCL exchange step (broadcast + per-LoRA-block all-reduces from inside the backward) on one GPU.

--config 1 (default; BASELINE.json configs[1], and configs[2] under --gpus 8): sam2.1_hiera_base_plus + LoRA_qkv r=4,
    bf16 fwd+bwd, 16 frames of 1024^2 per GPU.  One step = SAM2Base.forward_image (sam2_base.py:467-479: trunk + FPN neck +
    conv_s0 / conv_s1 on the two high-resolution levels -- the encoder call of SAM2Train.forward, training/model/sam2.py:125,
    which is what train_lora_sam.py / train_cl.py drive), loss = sum_l mean(fpn_l^2) over its three outputs, backward to the
    LoRA A/B factors, mean all-reduce of the LoRA gradients (N>1), AdamW step on them.  --plain-encoder times
    ImageEncoder.forward alone (256-channel levels 0 / 1, no conv_s0 / conv_s1) instead.
--config 3 (configs[3]): sam2.1_hiera_large + LoRA_qkv r=8, bf16 encoder-only inference (trunk + neck under no_grad),
    64 frames per GPU.  One step = one batch.
--config 4 (configs[4]): train_cl_kmean-style feature extraction: Hiera-B+ + LoRA r=4, Hiera.get_embedding + per-frame
    spatial mean of the stage-4 map (896-d), 10,000 frames sharded by frame over the ranks (features.shard_range), batches
    of 64.  One step = one batch; without --steps the pass covers all 10,000 frames.
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
import types

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0    # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0        # HBM3E spec (same table; ~6.3 TB/s is what a streaming copy achieves)
BPLUS = dict(embed_dim=112, num_heads=2)
LARGE = dict(embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43), window_spec=(8, 4, 16, 8))
# algorithmic FLOPs per 1024^2 frame
GF_BPLUS_FWD_BWD = 657.4e9   # Hiera-B+ r=4 forward + LoRA backward, neck lateral convs as the reference executes them
# forward_image composes conv_s0 / conv_s1 into the level-0 / 1 lateral convs (sam2_base.py:473-478 after
# image_encoder.py:112-114): 2*65536*112*32 + 2*16384*224*64 = 0.94 GF instead of 2*65536*112*256 + 2*16384*224*256 = 5.64 GF,
# and the reference's own conv_s0 / conv_s1 (2*65536*256*32 + 2*16384*256*64 = 1.61 GF, outside SURVEY's 657.4) never run.
# The whole-step fraction is priced on what this implementation must compute: 657.4 - 5.64 + 0.94.
GF_BPLUS_FWD_BWD_COMPOSED = GF_BPLUS_FWD_BWD - 5.64e9 + 0.94e9
GF_BPLUS_FWD = 621.3e9       # forward incl. neck (7.05 GF); the trunk-only extraction pass is 614.2
GF_BPLUS_TRUNK = 614.2e9
GF_LARGE_FWD = 1821.4e9      # Hiera-L r=8 forward incl. neck

# live per-launch records of liblsam (lsam_prof_arm/read): cfg id -> (name, unit of `work`)
KERNEL_NAMES = {
    9448: ("tailpair<14,28> stage-3 block tail (proj+res+LN2+MLP+res+next LN1, 448 wide)", "flop"),
    9224: ("tail<7,14> stage-2 block tail (224 wide)", "flop"),
    2224: ("gemm3 256x224 persistent bf16 GEMM, bf16 out", "flop"),
    2225: ("gemm3 256x224 persistent bf16 GEMM, fp32 out", "flop"),
    5004: ("qkvattn4: qkv (k | v) projection + 4x4-window attention in one kernel (stage 2)", "flop"),
    2016: ("streamlin: small-K Linear, weights resident in LDS, rows streamed", "flop"),
    2226: ("gemm3 256x224 persistent bf16 GEMM, column-guarded (N % 224 != 0), bf16 out", "flop"),
    2227: ("gemm3 256x224 persistent bf16 GEMM, column-guarded (N % 224 != 0), fp32 out", "flop"),
    2112: ("gemm2 256x112-tile bf16 GEMM, bf16 out (N % 224 != 0)", "flop"),
    2113: ("gemm2 256x112-tile bf16 GEMM, fp32 out", "flop"),
    2032: ("skinny32: streaming N = 32 product (LoRA u / t columns)", "flop"),
    2128: ("gemm_cfg 128-wide-tile bf16 GEMM (N % 112 != 0: Hiera-L widths, LoRA u columns)", "flop"),
    4128: ("gemm_cfg 128-wide-tile fp32 GEMM", "flop"),
    5000: ("global attention (attn2_fwd)", "flop"),
    5001: ("window attention (attnw / attn2 / attn_fwd)", "flop"),
    5002: ("attention backward (LoRA blocks)", "flop"),
    5003: ("kvattn: k|v projection + 8x8-window attention of the q-pool block 2, one kernel", "flop"),
    100: ("ln_fwd", "byte"), 101: ("ln_bwd", "byte"), 102: ("cast fp32->bf16", "byte"),
    103: ("lateral 1x1 conv (levels 0/1)", "byte"), 104: ("patch_embed 7x7/4 conv + pos", "byte"),
    105: ("winblock stage-1 attention half", "byte"), 106: ("mlpblock stage-1 MLP half", "byte"),
    107: ("maxpool2x2", "byte"), 108: ("spatial_mean", "byte"), 109: ("LoRA dA/dB token reductions", "byte"),
    110: ("neck level (lateral + top-down add + NCHW store)", "byte"),
}


def measured_traffic(kernel_key: str):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/<round>/traffic.json, written by
    tools/pmc_traffic.py from separate FETCH_SIZE and WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950).  bench.py cannot run the counter passes on itself; null when absent."""
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
            j = json.load(fh)
    except (OSError, ValueError):
        return None
    per = j.get("per_kernel", {})
    if kernel_key in per:
        return per[kernel_key]
    return j.get("hbm_bytes_per_launch") if kernel_key == j.get("kernel_key", "tailpair") else None


def measured_trace():
    """Kernel time per step of the committed rocprofv3 kernel trace of this command (profiles/<round>/bench_n1_kernel_summary.txt,
    written by tools/prof_summary.py over the TIMED steps only): the figure that reconciles with `ms_per_step`.  The live
    event brackets below cannot: every bracket adds two event packets to the stream (the bracketed steps run ~12 % slower
    than the timed ones), so their sum over-states the kernels even net of the empty-bracket cost."""
    import re
    root = os.path.join(REPO, "profiles")
    best = None
    if os.path.isdir(root):
        for r in sorted(os.listdir(root)):
            f = os.path.join(root, r, "bench_n1_kernel_summary.txt")
            if os.path.isfile(f):
                best = f
    if best is None:
        return None
    try:
        text = open(best).read()
    except OSError:
        return None
    m = re.search(r"kernel durations ([\d.]+) ms/step, of which liblsam ([\d.]+) and torch[^\d]*([\d.]+)", text)
    if not m:
        return None
    out = {"file": os.path.relpath(best, REPO), "all_kernels_ms_per_step": float(m.group(1)), "liblsam_ms_per_step": float(m.group(2)),
           "torch_ms_per_step": float(m.group(3))}
    d = re.search(r"avg\s+([\d.]+) us\s+\('void lsam::tailpair_kernel<14, 28", text)   # the roofline kernel in the same trace
    if d:
        out["tailpair_avg_launch_us"] = float(d.group(1))
    return out


def build(device, cfg=BPLUS, lora_rank=4):
    import image_segmentation_amd as m
    from image_segmentation_amd.synth import fill_synthetic_, synth_tensor
    enc = m.build_image_encoder(dict(cfg))
    fill_synthetic_(enc, 0)
    m.wrap_image_encoder_lora(types.SimpleNamespace(image_encoder=enc), lora_rank)
    with torch.no_grad():
        for name, p in enc.named_parameters():
            if "linear_a_" in name or "linear_b_" in name:
                p.copy_(synth_tensor(name, tuple(p.shape), 0))
    enc = enc.to(device)
    enc.trunk.compute_dtype = torch.bfloat16
    return enc


class _MeanSquare(torch.autograd.Function):
    """mean(f^2) with fp32 accumulation as ONE reduction over f (||f||_2^2 / numel: no fp32 copy or square of the level is
    materialised) and ONE elementwise kernel in the backward (f * (2 g / numel), the scale an fp32 scalar, the product
    rounded once to f's dtype) -- torch's own graph for the same expression differentiates pow and norm separately, two
    passes over the level."""

    @staticmethod
    def forward(ctx, f):
        ctx.save_for_backward(f)
        return torch.linalg.vector_norm(f, dtype=torch.float32) ** 2 / f.numel()

    @staticmethod
    def backward(ctx, g):
        (f,) = ctx.saved_tensors
        return f * (g.float() * (2.0 / f.numel()))


def fpn_loss(out):
    """sum_l mean(fpn_l^2): the synthetic loss of SURVEY.md section 8d."""
    return sum(_MeanSquare.apply(f) for f in out["backbone_fpn"])


def cpu_baseline(config: int, warm: int, iters: int):
    """The oracle (CPU restatement of the reference path, pinned to reference goldens; kind "port") timed on this box's
    host cores on a bounded sample of the same workload: batch 1, fp32, one 1024^2 frame per iteration."""
    from image_segmentation_amd.synth import synth_frames
    from oracle import hiera_oracle as O
    from oracle.params import make_params
    cores = torch.get_num_threads()      # the box's OMP_NUM_THREADS share (16 on a 1-GPU box), not the machine's CPU count
    cfg, rank = (LARGE, 8) if config == 3 else (BPLUS, 4)
    spec = O.block_specs(**cfg)
    p = make_params(spec, rank, 0)
    img = synth_frames(1, 1024, 0)
    tp = {k[len("trunk."):]: v for k, v in p.items() if k.startswith("trunk.")}

    if config == 1:
        what = "fwd+bwd (configs[0])"

        def it():
            for t in p.values():
                t.grad = None
            O.fpn_loss(O.image_encoder_forward(p, img, spec)).backward()
    elif config == 3:
        what = "encoder forward, no grad"

        def it():
            with torch.no_grad():
                O.image_encoder_forward(p, img, spec)
    else:
        what = "get_embedding + spatial mean, no grad"

        def it():
            with torch.no_grad():
                O.trunk_forward(tp, img, spec, n_grad_blocks=0)[-1].mean(dim=(2, 3))

    for _ in range(warm):
        it()
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    dt = time.perf_counter() - t0
    name = "Hiera-L r=8" if config == 3 else "Hiera-B+ r=4"
    return {"value": round(iters / dt, 4), "unit": "frames/s", "cores": cores, "host_cpu_count": os.cpu_count(),
            "kind": "port",
            "sample": f"{iters} timed iterations ({warm} warm-up) of {name} {what}, batch 1, fp32, one 1024x1024 frame, "
                      f"torch CPU kernels on {cores} threads"}


def kernel_table(psteps: int):
    """Per-kernel-class live timing (HIP events on the launch stream, recorded inside liblsam) of `psteps` steps."""
    from image_segmentation_amd import _C
    cap = 600 * psteps
    ms = (C.c_float * cap)()
    wk = (C.c_double * cap)()
    cfg = (C.c_int * cap)()
    n = _C.lib().lsam_prof_read(cap, ms, wk, cfg, None)
    agg = {}
    for i in range(n):
        a = agg.setdefault(cfg[i], [0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms[i]
        a[2] += wk[i]
    rows = []
    for c, (cnt, tms, twk) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        name, unit = KERNEL_NAMES.get(c, (f"cfg {c}", "flop" if c >= 1000 else "byte"))
        rate = twk / max(tms * 1e-3, 1e-12)
        row = {"cfg": c, "kernel": name, "launches_per_step": round(cnt / psteps, 2), "ms_per_step": round(tms / psteps, 3),
               "avg_launch_us": round(1e3 * tms / cnt, 2)}
        if unit == "flop":
            row.update(achieved=round(rate / 1e12, 1), unit="TFLOP/s", frac=round(rate / 1e12 / PEAK_BF16_TFLOPS, 4))
        else:
            row.update(achieved=round(rate / 1e9, 1), unit="GB/s", frac=round(rate / 1e9 / PEAK_HBM_GBS, 4))
        rows.append(row)
    return rows, agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=1, choices=(1, 3, 4), help="BASELINE.json configs[] index")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU and step")
    ap.add_argument("--frames", type=int, default=10000, help="config 4: frames in the whole pass")
    ap.add_argument("--cpu-iters", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plain-encoder", action="store_true", help="config 1: ImageEncoder.forward instead of forward_image")
    ap.add_argument("--heads", action="store_true",
                    help="config 1, informational (not BASELINE.json's metric): the whole prompted-frame step -- forward_image, SAM "
                         "prompt encoder + mask decoder (fp32 operands), MultiStepMultiMasksAndIous, backward through all of it")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ      # under torch.distributed.run
    if not launched and args.gpus > 1:
        # Plain `python bench.py --gpus N`: start the N workers ourselves, as a child process, BEFORE this process makes any
        # GPU call (importing torch makes none); the child's rank 0 prints the line on the stdout we share.
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if launched:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=device)    # backend "nccl" IS RCCL on ROCm (train_lora_sam.py:32)
    grouped = launched

    from image_segmentation_amd import _C
    from image_segmentation_amd.dp import LoRAGradReducer
    from image_segmentation_amd.features import extract_batch, shard_range
    from image_segmentation_amd.synth import synth_frames_device

    config = args.config
    B = args.batch or {1: 16, 3: 64, 4: 64}[config]
    warmup = args.warmup if args.warmup is not None else {1: 5, 3: 2, 4: 3}[config]

    if config == 1:
        enc = build(device)
        lora = [p for p in enc.parameters() if p.requires_grad]
        if not args.plain_encoder:
            from image_segmentation_amd.sam_heads import SAM2Base
            torch.manual_seed(0)
            sam = SAM2Base(image_encoder=enc).to(device)       # heads at their seeded default init: only conv_s0 / conv_s1 run
            for n_, p_ in sam.named_parameters():
                if not n_.startswith("image_encoder."):
                    p_.requires_grad = False                   # train_lora_sam.py:116-117 freezes everything but LoRA
        reducer = LoRAGradReducer(lora).attach(enc.trunk)   # per-block buckets, all-reduce overlapped with backward
        reducer.broadcast_params(0)
        opt = torch.optim.AdamW(lora, lr=1e-4, fused=True)   # config.yaml:6 of the reference; fused: the same update as ONE kernel
        img = synth_frames_device(B, 1024, 0, device, first_index=rank * B)   # resident before timing
        steps = args.steps if args.steps is not None else 400   # ~6.6 s timed: longer than a 5-s activity sampler's period, and sustained clocks
        frames_per_step = B
        gf_per_frame = GF_BPLUS_FWD_BWD if args.plain_encoder else GF_BPLUS_FWD_BWD_COMPOSED

        if args.heads:
            from image_segmentation_amd.loss_fns import CORE_LOSS_KEY, MultiStepMultiMasksAndIous
            crit = MultiStepMultiMasksAndIous({"loss_mask": 20, "loss_dice": 1, "loss_iou": 1, "loss_class": 1}, supervise_all_iou=True,
                                              iou_use_l1_loss=True, pred_obj_scores=True)       # config.yaml:26-37
            yy, xx = torch.meshgrid(torch.arange(1024, device=device), torch.arange(1024, device=device), indexing="ij")
            tgt = (((yy - 512) ** 2 + (xx - 512) ** 2) < 300 ** 2).float().expand(1, B, 1024, 1024).contiguous()
            clicks = torch.full((B, 1, 2), 512.0, device=device)
            clabels = torch.ones(B, 1, dtype=torch.int32, device=device)

        def step(i):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = enc(img) if args.plain_encoder else sam.forward_image(img)
            if args.heads:
                o = sam.forward_first_frame(out, clicks, clabels)
                loss = crit([{"multistep_pred_multimasks_high_res": [o["high_res_multimasks"]], "multistep_pred_ious": [o["ious"]],
                              "multistep_object_score_logits": [o["object_score_logits"]]}], tgt)[CORE_LOSS_KEY]
            else:
                loss = fpn_loss(out)
            loss.backward()
            reducer.reduce()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss
    elif config == 3:
        enc = build(device, LARGE, 8)
        img = synth_frames_device(B, 1024, 0, device, first_index=rank * B)
        steps = args.steps if args.steps is not None else 25    # ~5.5 s timed
        frames_per_step = B
        gf_per_frame = GF_LARGE_FWD

        def step(i):
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                out = enc(img)
            return out["vision_features"].float().abs().max()
    else:
        enc = build(device)
        lo, hi = shard_range(args.frames, rank, world)       # this rank's slice of the 10,000 frames: no data-path collective
        pool_n = 4 * B                                        # resident pool of distinct frames, cycled through the pass
        pool = synth_frames_device(pool_n, 1024, 0, device, first_index=rank * pool_n)
        steps = args.steps if args.steps is not None else (hi - lo + B - 1) // B
        frames_per_step = B
        gf_per_frame = GF_BPLUS_TRUNK
        feats = torch.empty(steps * B, enc.trunk.channel_list[0], dtype=torch.float32, device=device)

        def source(first, n):
            o = first % pool_n
            return pool[o:o + n] if o + n <= pool_n else torch.cat([pool[o:], pool[: n - (pool_n - o)]])

        def step(i):
            f = extract_batch(enc.trunk, source(lo + i * B, B))
            feats[i * B:(i + 1) * B] = f
            return f[0, 0]

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        last = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    collectives = dict(reducer.stats) if config == 1 else None      # issued during warm-up + timed steps
    assert torch.isfinite(last).item(), "non-finite result"
    fps = world * frames_per_step * steps / elapsed

    roofline = None
    kernels = None
    if rank == 0:
        # live per-launch timing of every liblsam kernel class: HIP events on the launch stream, recorded inside the
        # library, over `psteps` further steps of the same workload.  achieved = algorithmic FLOPs (or HBM bytes) of
        # those launches / their duration.
        psteps = min(2, steps)
        _C.check(_C.lib().lsam_prof_arm(600 * psteps))
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for i in range(psteps):
            step(i)
        torch.cuda.synchronize()
        prof_wall_ms = 1e3 * (time.perf_counter() - tp0) / psteps
        kernels, agg = kernel_table(psteps)
        bracket = C.c_float(0.0)
        _C.check(_C.lib().lsam_prof_bracket_cost(_C.stream(), 200, C.byref(bracket)))
        n_launch = sum(a[0] for a in agg.values()) / psteps
        mfma_rows = [r for r in kernels if r["unit"] == "TFLOP/s"]
        dom = max(mfma_rows, key=lambda r: r["ms_per_step"])
        cnt, tms, twk = agg[dom["cfg"]]
        traffic_key = {9448: "tailpair", 9224: "tail224", 2224: "gemm3"}.get(dom["cfg"], str(dom["cfg"]))
        roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": dom["frac"], "traffic": measured_traffic(traffic_key),
                    "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_launch_us"],
                    "avg_launch_gflop": round(twk / cnt / 1e9, 3), "kernel_ms_per_step": dom["ms_per_step"],
                    # Sum of every liblsam launch's event bracket.  A bracket spans the kernel plus the stream's cost of the two
                    # event packets, so the raw sum can exceed a step's wall time; `empty_bracket_us` (lsam_prof_bracket_cost:
                    # the same event pair around nothing) is that cost, and the `_net` sum has it subtracted per launch.  The
                    # bracketed steps themselves run slower than the timed ones (`profiled_step_wall_ms` vs `ms_per_step`).
                    "all_timed_kernel_ms_per_step": round(sum(r["ms_per_step"] for r in kernels), 3),
                    "empty_bracket_us": round(1e3 * bracket.value, 3), "timed_launches_per_step": round(n_launch, 1),
                    "all_timed_kernel_ms_per_step_net": round(sum(r["ms_per_step"] for r in kernels) - n_launch * bracket.value, 3),
                    "profiled_step_wall_ms": round(prof_wall_ms, 3),
                    "kernel_trace": measured_trace() if config == 1 and not (args.plain_encoder or args.heads) else None}
        if dom["cfg"] == 9448:
            roofline["algorithmic_hbm_bytes_per_launch"] = 12 * B * 64 * 64 * 448
    if grouped:
        dist.barrier()

    if rank == 0:
        model = "sam2.1_hiera_large + LoRA_qkv r=8" if config == 3 else "sam2.1_hiera_base_plus + LoRA_qkv r=4"
        what = {1: ("bf16 fwd+bwd of ImageEncoder.forward" if args.plain_encoder else
                    "bf16 fwd+bwd of SAM2Base.forward_image (trunk + FPN neck + conv_s0/conv_s1)")
                   + (" + SAM heads (fp32) + MultiStepMultiMasksAndIous, fwd+bwd [informational: not BASELINE.json's metric]" if args.heads else "")
                   + " (+LoRA-grad all-reduce, AdamW)",
                3: "bf16 encoder-only inference (trunk + FPN neck, no grad)",
                4: f"bf16 feature extraction (get_embedding + per-frame spatial mean), {steps * B * world} frames "
                   f"sharded by frame"}[config]
        metric = {1: "frames/sec LoRA-SAM2 Hiera-B+ fwd+bwd @1024^2",
                  3: "frames/sec LoRA-SAM2 Hiera-L encoder-only inference @1024^2",
                  4: "frames/sec LoRA-SAM2 Hiera-B+ feature-extraction pass @1024^2"}[config]
        line = {
            "metric": metric, "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic N(0,1) frames generated on-device before the timed region"
                    + (f" (a resident pool of {4 * B} distinct frames cycled through the pass)" if config == 4 else "")
                    + "; seeded synthetic weights (no checkpoints offline)",
            "config": {"workload": f"BASELINE.json configs[{config}]: {model}, {what}, 1024x1024",
                       "frames_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "residual_stream": "fp32"},
            "mfma_frac_whole_step": round(fps / world * gf_per_frame / (PEAK_BF16_TFLOPS * 1e12), 4),
            "gflop_per_frame": round(gf_per_frame / 1e9, 2),
            "process_group": ("nccl (RCCL), %d rank(s)" % world) if grouped else None,
            "collectives_issued": collectives,
            "roofline": roofline,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            iters = args.cpu_iters if args.cpu_iters is not None else {1: 10, 3: 4, 4: 10}[config]
            line["cpu_baseline"] = cpu_baseline(config, 3 if config != 3 else 1, iters)
        print(json.dumps(line), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
