"""MI355X-native LoRA-SAM2 Hiera image-encoder hot path (drop-in for the reference's
`LoRA_qkv` / `image_encoder` module API).  Arithmetic lives in csrc/ (HIP, gfx950) behind
the C ABI of include/lsam.h; these modules are the host-side mirror of the reference API."""
from .hieradet import Hiera, MultiScaleAttention, MultiScaleBlock, PatchEmbed, MLP  # noqa: F401
from .image_encoder import FpnNeck, ImageEncoder, PositionEmbeddingSine, build_image_encoder, forward_image  # noqa: F401
from .lora_qkv import (LoRA_qkv, LoRABank, custom_load_lora_parameters, custom_save_lora_parameters,  # noqa: F401
                       wrap_image_encoder_lora)
from .sam_heads import (Attention, Attention_LoRA, MaskDecoder, PromptEncoder, SAM2Base, TwoWayAttentionBlock,  # noqa: F401
                        TwoWayTransformer, wrap_decoder_lora)
from .loss_fns import MultiStepMultiMasksAndIous, dice_loss, iou_loss, sigmoid_focal_loss  # noqa: F401
