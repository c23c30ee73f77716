"""Attention backend selector, API-compatible with the reference enum (src/UCF_VIT/utils/fused_attn.py:13-17).

In this implementation every member maps to the same hand-written gfx950 fused-attention kernel
(ucfvit_attention_fwd/bwd): the member only records what the caller asked for, so reference scripts and
configs that pass FLASH / CK / DEFAULT / NONE keep working.  `HIP` names the native kernel explicitly.
"""
import enum


class FusedAttn(enum.Enum):
    FLASH = "FLASH"      # reference: xformers FlashAttention op
    CK = "CK"            # reference: xformers Composable-Kernel op
    DEFAULT = "DEFAULT"  # reference: F.scaled_dot_product_attention
    NONE = "NONE"        # reference: explicit q@k^T / softmax / @v
    HIP = "HIP"          # this build: libucfvit_hip.so fused MFMA attention (what all members run)
