"""Dev aid: is one whole training step capturable in a HIP graph (torch.cuda.CUDAGraph), and what does replay gain?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from image_segmentation_amd.synth import synth_frames_device
dev = torch.device("cuda", 0)
enc = bench.build(dev)
lora = [p for p in enc.parameters() if p.requires_grad]
opt = torch.optim.AdamW(lora, lr=1e-4, capturable=True)
img = synth_frames_device(16, 1024, 0, dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = enc(img)
    loss = bench.fpn_loss(out)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=False)
    return loss
def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
print("eager ms/step", round(timeit(step), 3), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
print("captured", flush=True)
print("graph ms/step", round(timeit(g.replay), 3), "loss", float(static_loss), flush=True)
l0 = float(static_loss); g.replay(); torch.cuda.synchronize(); print("loss after another replay", float(static_loss), "(changes: optimizer steps are applied)")
