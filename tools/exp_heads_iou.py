import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import BPLUS, load_golden
from test_masks_cpu import golden_masks
from test_trunk_gpu import make_encoder
from image_segmentation_amd.sam_heads import SAM2Base
from image_segmentation_amd.synth import synth_frames
from oracle import sam_heads as SH
import torch.nn.functional as F
g = load_golden("masks_bplus_r4"); n = g["coords"].shape[0]
enc = make_encoder(BPLUS)
model = SAM2Base(image_encoder=enc); model.load_state_dict(SH.make_head_params(0), strict=False); model = model.cuda()
hi, lo = golden_masks(g, "high_bits", 1024), golden_masks(g, "low_bits", 256)
coords, labels = torch.from_numpy(g["coords"]).cuda(), torch.from_numpy(g["labels"]).cuda()
dec = model.sam_mask_decoder
def report(tag, fpn):
    with torch.no_grad():
        s0 = F.conv2d(fpn[0].float(), dec.conv_s0.weight, dec.conv_s0.bias)
        s1 = F.conv2d(fpn[1].float(), dec.conv_s1.weight, dec.conv_s1.bias)
    out = model.forward_first_frame({"backbone_fpn": [s0, s1, fpn[2].float()]}, coords, labels)
    ious = [SH.mask_iou(out["high_res_masks"][i].cpu() > 0, hi[i]) for i in range(n)]
    ious_lo = [SH.mask_iou(out["low_res_masks"][i].cpu() > 0, lo[i]) for i in range(n)]
    print(tag, ious, ious_lo, flush=True)
img = synth_frames(n, 1024, 0, device="cuda")
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
with torch.no_grad():
    enc.trunk.compute_dtype = torch.float32
    f32 = enc(img)["backbone_fpn"]
    enc.trunk.compute_dtype = torch.bfloat16
    fa = enc(img)["backbone_fpn"]                      # bf16 trunk, fp32 neck
    with torch.autocast("cuda", dtype=torch.bfloat16):
        fb = enc(img)["backbone_fpn"]                  # bf16 trunk, bf16 neck
report("fp32", f32); report("bf16 trunk + fp32 neck", fa); report("bf16 trunk + bf16 neck", fb)
print("rel err vs fp32, fp32 neck:", [rel(a, b) for a, b in zip(fa, f32)])
print("rel err vs fp32, bf16 neck:", [rel(a, b) for a, b in zip(fb, f32)])
print("bf16 neck vs fp32 neck:", [rel(a, b) for a, b in zip(fb, fa)])
report("mix: fb0 fb1 fa2", [fb[0], fb[1], fa[2]]); report("mix: fa0 fa1 fb2", [fa[0], fa[1], fb[2]])
report("fa rounded to bf16", [t.bfloat16() for t in fa])
