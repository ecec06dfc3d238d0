"""Dev aid: does running the 16-frame step as two 8-frame half-batches on two HIP streams (kernel boundaries of one stream
under the other's kernels) beat one 16-frame stream?  Same global batch, gradients accumulate (mean of the halves)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from image_segmentation_amd.sam_heads import SAM2Base
from image_segmentation_amd.synth import synth_frames_device

dev = torch.device("cuda", 0)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
img = synth_frames_device(16, 1024, 0, dev)


def make():
    enc = B.build(dev)
    sam = SAM2Base(image_encoder=enc).to(dev)
    for n, p in sam.named_parameters():
        if not n.startswith("image_encoder."):
            p.requires_grad = False
    return enc, sam


def run(label, step, steps=20, warm=5):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {16 * steps / dt:8.1f} frames/s  {1e3 * dt / steps:7.3f} ms/step", flush=True)


enc, sam = make()
lora = [p for p in enc.parameters() if p.requires_grad]
opt = torch.optim.AdamW(lora, lr=1e-4, fused=True)


def one():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = sam.forward_image(img)
    B.fpn_loss(out).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


run("1 stream x 16 frames", one)

# NS independent model replicas sharing the SAME LoRA parameters would need one plan per stream; the trunk keeps one
# workspace per module, so each stream gets its own module instance whose LoRA parameters alias the first one's.
mods = [(enc, sam)] + [make() for _ in range(NS - 1)]
for e2, _ in mods[1:]:
    for (n1, p1), (n2, p2) in zip(enc.named_parameters(), e2.named_parameters()):
        if p1.requires_grad:
            p2.data = p1.data
streams = [torch.cuda.Stream() for _ in range(NS)]
chunk = 16 // NS


def multi():
    cur = torch.cuda.current_stream()
    for i, (st, (e_, s_)) in enumerate(zip(streams, mods)):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = s_.forward_image(img[i * chunk:(i + 1) * chunk])
            (B.fpn_loss(out) / NS).backward()
    for st in streams:
        cur.wait_stream(st)
    for e2, _ in mods[1:]:
        for p1, p2 in zip(lora, [p for p in e2.parameters() if p.requires_grad]):
            p1.grad.add_(p2.grad)
            p2.grad = None
    opt.step()
    opt.zero_grad(set_to_none=True)
    for e2, _ in mods[1:]:
        e2.trunk.invalidate_packed() if hasattr(e2.trunk, "invalidate_packed") else None


run(f"{NS} streams x {chunk} frames", multi)
run("1 stream x 16 frames (again)", one)
