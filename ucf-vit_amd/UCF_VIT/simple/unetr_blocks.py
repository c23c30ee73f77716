"""UNETR convolutional decoder blocks (3x3(x3) conv + instance norm + LeakyReLU residual blocks, transposed-conv upsampling,
1x1 output conv).  In the reference these come from the un-vendored dependency monai>=1.4.0 (UnetrBasicBlock, UnetrPrUpBlock,
UnetrUpBlock, UnetOutBlock; src/UCF_VIT/simple/arch.py:33-34,808-940): the classes below restate that published block
structure and its state_dict naming so checkpoints line up.  PARITY UNPINNED: monai is not installed in the build container
and the reference ships no fixtures for it.

Two execution paths over the SAME parameters (nn.Conv3d / nn.ConvTranspose3d modules are kept as the parameter containers):
  forward_cl(...)  the 3-D decoder on the HIP kernels end to end: channels-last bf16 activations [B, X, Y, Z, C],
                   3x3x3 convolutions as implicit GEMMs on MFMA, transposed convolutions as GEMM + depth-to-space, instance norm + LeakyReLU
                   (+ residual) fused, csrc/conv3d.hip + csrc/unetr_decoder.hip through UCF_VIT/_hip/conv.py.  UNETR uses it whenever
                   hip_decoder_supported() holds (3-D, kernel 3 / stride 1 / upsample 2, channel counts the kernels tile).
  forward(...)     N C (D) H W tensors: torch's own convolutions with the fused HIP instance-norm kernels between them — ONLY for models
                   built with UNETR(allow_torch_decoder=True) (2-D, unusual channel counts); without that opt-in UNETR.forward raises
                   instead of dropping to a second backend.
"""
import torch
import torch.nn as nn

from UCF_VIT._hip import conv as HC
from UCF_VIT._hip import functional as HF


def _conv(nd, cin, cout, k, s, transposed=False, bias=False):
    cls = {(2, False): nn.Conv2d, (3, False): nn.Conv3d, (2, True): nn.ConvTranspose2d, (3, True): nn.ConvTranspose3d}[(nd, transposed)]
    pad = 0 if transposed else (k - s + 1) // 2
    m = nn.Sequential()
    m.add_module("conv", cls(cin, cout, kernel_size=k, stride=s, padding=pad, bias=bias))   # key '<name>.conv.weight' like monai's Convolution
    return m


def _inorm(nd, c):
    return (nn.InstanceNorm2d if nd == 2 else nn.InstanceNorm3d)(c)


class UnetResBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name="instance"):
        super().__init__()
        nd = spatial_dims
        self.conv1 = _conv(nd, in_channels, out_channels, kernel_size, stride)
        self.conv2 = _conv(nd, out_channels, out_channels, kernel_size, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.01, inplace=True)
        self.norm1, self.norm2 = _inorm(nd, out_channels), _inorm(nd, out_channels)
        self.downsample = in_channels != out_channels or stride != 1
        if self.downsample:
            self.conv3 = _conv(nd, in_channels, out_channels, 1, stride)
            self.norm3 = _inorm(nd, out_channels)

    def forward(self, inp):
        if not inp.is_cuda:
            raise RuntimeError("UnetResBlock: the normalisation / activation kernels run on the MI355X only; there is no CPU path")
        # convolutions on MIOpen; every instance norm with what follows it (LeakyReLU, residual add + LeakyReLU) is ONE fused HIP op
        ns = self.lrelu.negative_slope
        out = HF.instnorm_act(self.conv1(inp), None, self.norm1.eps, ns)
        out = self.conv2(out)
        residual = HF.instnorm_act(self.conv3(inp), None, self.norm3.eps, 1.0) if self.downsample else inp
        return HF.instnorm_act(out, residual, self.norm2.eps, ns)

    def forward_cl(self, inp):
        """inp channels-last bf16 [B, X, Y, Z, Cin] -> [B, X, Y, Z, Cout]: the whole block as one autograd node (_hip/conv.py:UnetResBlockFn)"""
        if self.norm1.eps != self.norm2.eps or (self.downsample and self.norm3.eps != self.norm1.eps):
            raise ValueError("UnetResBlock: the fused block takes one eps for its normalisations")
        return HC.unet_res_block(inp, self.conv1.conv.weight, self.conv2.conv.weight, self.conv3.conv.weight if self.downsample else None,
                                 self.norm1.eps, self.lrelu.negative_slope)


class UnetrBasicBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, res_block=False):
        super().__init__()
        self.layer = UnetResBlock(spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name)

    def forward(self, inp):
        return self.layer(inp.float())

    def forward_cl(self, inp):
        return self.layer.forward_cl(inp)


class UnetrPrUpBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, num_layer, kernel_size, stride, upsample_kernel_size, norm_name,
                 conv_block=False, res_block=False):
        super().__init__()
        u = upsample_kernel_size
        self.transp_conv_init = _conv(spatial_dims, in_channels, out_channels, u, u, transposed=True)
        self.blocks = nn.ModuleList([
            nn.Sequential(_conv(spatial_dims, out_channels, out_channels, u, u, transposed=True),
                          UnetResBlock(spatial_dims, out_channels, out_channels, kernel_size, stride, norm_name))
            for _ in range(num_layer)])

    def forward(self, x):
        x = self.transp_conv_init(x.float())
        for blk in self.blocks:
            x = blk(x)
        return x

    def forward_cl(self, x):
        x = HC.tconv2x2x2(x, self.transp_conv_init.conv.weight)
        for blk in self.blocks:
            x = blk[1].forward_cl(HC.tconv2x2x2(x, blk[0].conv.weight))
        return x


class UnetrUpBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size, norm_name, res_block=False):
        super().__init__()
        u = upsample_kernel_size
        self.transp_conv = _conv(spatial_dims, in_channels, out_channels, u, u, transposed=True)
        self.conv_block = UnetResBlock(spatial_dims, out_channels + out_channels, out_channels, kernel_size, 1, norm_name)

    def forward(self, inp, skip):
        out = self.transp_conv(inp.float())
        return self.conv_block(torch.cat((out, skip), dim=1))

    def forward_cl(self, inp, skip):
        return self.conv_block.forward_cl(HC.tconv2x2x2(inp, self.transp_conv.conv.weight, skip))      # (up-sampled, skip) concatenated in place


class UnetOutBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels):
        super().__init__()
        self.conv = _conv(spatial_dims, in_channels, out_channels, 1, 1, bias=True)

    def forward(self, inp):
        return self.conv(inp)

    def forward_cl(self, inp):
        """-> fp32 logits [B, X, Y, Z, n]: a view of a buffer whose voxel rows are padded to 8 columns (ops.dice_ce reads it in place)"""
        return HC.conv1x1x1(inp, self.conv.conv.weight, self.conv.conv.bias, out_fp32=True)


def hip_decoder_supported(spatial_dims, in_chans, embed_dim, feature_size, kernel_size=3, upsample_kernel_size=2):
    """can the whole skip-connection decoder run on csrc/conv3d.hip?  3-D, 3x3x3 / 2x2x2 kernels, an input of <= 8 channels (zero-padded to the
    8-channel MFMA operand), feature_size a multiple of 16 whose multiples (x2, x4, x8, and the concatenations x2 .. x16) the kernels tile
    (16, or any multiple of 32), embed_dim one of the input widths the transposed-convolution / pointwise kernels tile (8, 16, or a
    multiple of 32: _hip/conv.py conv3_cin_supported — 72 or 120 would pass a plain "% 8" rule and then fail inside forward)"""
    fs = feature_size
    return (spatial_dims == 3 and kernel_size == 3 and upsample_kernel_size == 2 and 1 <= in_chans <= 8
            and (embed_dim in (8, 16) or (embed_dim > 0 and embed_dim % 32 == 0))
            and fs % 16 == 0 and (fs == 16 or fs % 32 == 0) and (fs & (fs - 1)) == 0)
