"""North-star mask criterion: masks predicted from the HIP encoder's outputs reach IoU >= 0.99 against the
REFERENCE's masks (tests/golden/masks_bplus_r4.npz, produced by the reference's own SAM2Base on the same
synthetic frames and weights).  The SAM heads are not part of the hot path: they run as the CPU
restatement oracle/sam_heads.py (pinned to the reference in tests/test_masks_cpu.py) on the encoder
outputs copied back from the GPU."""
import pytest
import torch

from conftest import BPLUS
from test_masks_cpu import golden_masks
from test_trunk_gpu import make_encoder
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,bar", [(torch.float32, 0.999), (torch.bfloat16, 0.99)])
def test_mask_iou_vs_reference(dtype, bar):
    from image_segmentation_amd.synth import synth_frames
    from oracle import sam_heads as SH
    g = load_golden("masks_bplus_r4")
    n = g["coords"].shape[0]
    enc = make_encoder(BPLUS)
    enc.trunk.compute_dtype = dtype
    # bf16 = the benchmark's configuration: bf16 trunk AND the neck under bf16 autocast
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        fpn = enc(synth_frames(n, 1024, 0, device="cuda"))["backbone_fpn"]
    assert fpn[2].dtype == dtype
    out = SH.forward_sam_heads(SH.make_head_params(0), fpn, torch.from_numpy(g["coords"]), torch.from_numpy(g["labels"]))
    assert (out["best"].numpy() == g["best"]).all()
    assert (out["object_score_logits"] > 0).all()
    hi, lo = golden_masks(g, "high_bits", 1024), golden_masks(g, "low_bits", 256)
    ious = [SH.mask_iou(out["high_res_masks"][i] > 0, hi[i]) for i in range(n)]
    ious_lo = [SH.mask_iou(out["low_res_masks"][i] > 0, lo[i]) for i in range(n)]
    print(f"mask IoU vs reference ({dtype}): high-res {ious}, low-res {ious_lo}")
    assert min(ious) >= bar and min(ious_lo) >= bar, (ious, ious_lo)


@pytest.mark.parametrize("enc_dtype,head_dtype,bar", [(torch.float32, torch.float32, 0.999), (torch.bfloat16, torch.float32, 0.99),
                                                      (torch.bfloat16, torch.bfloat16, 0.97)])
def test_mask_iou_product_heads_vs_reference(enc_dtype, head_dtype, bar):
    """The same criterion with NOTHING on the CPU: frames -> SAM2Base.forward_image (HIP trunk, neck, composed conv_s0 /
    conv_s1) -> SAM2Base.forward_first_frame (HIP prompt encoder + mask decoder, image_segmentation_amd/sam_heads.py)
    against the reference's own masks.  The heads' default (and parity-grade) operand type is fp32; their bf16 mode on top of
    a bf16 encoder measures IoU 0.978-0.985 on this fixture, below the 0.99 bar, and is only checked not to fall apart."""
    from image_segmentation_amd.sam_heads import SAM2Base
    from image_segmentation_amd.synth import synth_frames
    from oracle import sam_heads as SH
    g = load_golden("masks_bplus_r4")
    n = g["coords"].shape[0]
    enc = make_encoder(BPLUS)
    enc.trunk.compute_dtype = enc_dtype
    model = SAM2Base(image_encoder=enc)
    model.load_state_dict(SH.make_head_params(0), strict=False)
    model = model.cuda()
    model.compute_dtype = head_dtype
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=enc_dtype == torch.bfloat16):
        out = model.forward_first_frame(model.forward_image(synth_frames(n, 1024, 0, device="cuda")),
                                        torch.from_numpy(g["coords"]).cuda(), torch.from_numpy(g["labels"]).cuda())
    assert (out["best"].cpu().numpy() == g["best"]).all()
    assert (out["object_score_logits"] > 0).all()
    hi, lo = golden_masks(g, "high_bits", 1024), golden_masks(g, "low_bits", 256)
    ious = [SH.mask_iou(out["high_res_masks"][i].cpu() > 0, hi[i]) for i in range(n)]
    ious_lo = [SH.mask_iou(out["low_res_masks"][i].cpu() > 0, lo[i]) for i in range(n)]
    print(f"product-path mask IoU vs reference (encoder {enc_dtype}, heads {head_dtype}): high-res {ious}, low-res {ious_lo}")
    assert min(ious) >= bar and min(ious_lo) >= bar, (ious, ious_lo)
