"""Adaptive patching on the GPU (reference: src/UCF_VIT/dataloaders/transform.py:9-55 `Patchify`, quadtree.py:84-174 `FixedQuadTree`).

The reference patches ONE image per call on the host: cv2 edge detection -> greedy fixed-length quadtree -> cv2 bicubic resize of every
leaf to patch_size x patch_size -> `np.reshape([S, p, p, C] -> [C, S, p*p])`.  This class does the tree and the resampling for a
whole batch on the MI355X (libucfvit_hip.so: ucfvit_quadtree_build / ucfvit_quadtree_serialize) and returns tensors in exactly the
layout VIT / MAE (adaptive_patching=True) consume.  Edge detection is NOT included (cv2.Canny has no counterpart here): the caller
passes the edge maps, e.g. computed by the reference's own transform on the host or by any detector that emits 0 / 255.
"""
import torch

from .._hip import ops


class Patchify(torch.nn.Module):
    """forward(img, edges) -> (seq_img, seq_size, seq_pos, nodes, count)
       img   fp32 [B, H, W, C] (channels last like the reference's numpy images; square images for square patches)
       edges uint8 [B, H, W], 0 / 255
       seq_img  fp32 [B, C, fixed_length, patch_size**2]      (the reference's per-image [C, S, p*p], transform.py:44-48)
       seq_size fp32 [B, fixed_length]    width of every leaf (0 = padding)
       seq_pos  fp32 [B, fixed_length, 2] leaf centres (x, y) ((-1, -1) = padding)
       nodes int32 [B, fixed_length, 4] = (x1, x2, y1, y2), count int32 [B]
    The training scripts feed the model `seq_ps = cat([seq_size[..., None], seq_pos], -1)` (train_class_simple.py:325-337)."""

    def __init__(self, fixed_length=196, patch_size=16, num_channels=3):
        super().__init__()
        if fixed_length % 3 != 1:
            raise ValueError("Quadtree fixed length needs to be 3n+1, where n is some integer")      # train_unetr_simple.py:214
        self.fixed_length, self.patch_size, self.num_channels = fixed_length, patch_size, num_channels

    @torch.no_grad()
    def forward(self, img, edges):
        if img.dim() != 4 or img.shape[-1] != self.num_channels:
            raise ValueError(f"Patchify: img must be [B, H, W, C={self.num_channels}], got {tuple(img.shape)}")
        img = img if img.dtype == torch.float32 else img.float()
        img = img if img.is_contiguous() else img.contiguous()
        edges = edges if edges.is_contiguous() else edges.contiguous()
        nodes, _, count, seq_ps = ops.quadtree_build(edges, self.fixed_length)
        seq_img = ops.quadtree_serialize(img, nodes, count, self.patch_size)
        return seq_img, seq_ps[..., 0], seq_ps[..., 1:], nodes, count


class Patchify_3D(torch.nn.Module):
    """3-D counterpart (reference transform.py:57-132 + octree.py:66-151): forward(vol, domain) -> (seq_img, seq_size, seq_pos, nodes, count)
       vol    fp32 [B, N, N, N, C] (cubic, channels last), domain uint8 [B, N, N, N] = the reference's `edges` volume
              (edge counter * norm_factor, norm_factor = int(255 / num_channels), transform.py:118-121)
       seq_img fp32 [B, C, fixed_length, patch_size**3], seq_size [B, fixed_length], seq_pos [B, fixed_length, 3]
       nodes int32 [B, fixed_length, 6] = (x1, x2, y1, y2, z1, z2); leaves are resampled with the reference's aligned-corner linear rule."""

    def __init__(self, fixed_length=729, patch_size=8, num_channels=1):
        super().__init__()
        if fixed_length % 7 != 1:
            raise ValueError("Octtree fixed length needs to be 7n+1, where n is some integer")       # train_unetr_simple.py:218
        self.fixed_length, self.patch_size, self.num_channels = fixed_length, patch_size, num_channels
        self.norm_factor = int(255 / num_channels)

    @torch.no_grad()
    def forward(self, vol, domain):
        if vol.dim() != 5 or vol.shape[-1] != self.num_channels:
            raise ValueError(f"Patchify_3D: vol must be [B, N, N, N, C={self.num_channels}], got {tuple(vol.shape)}")
        vol = vol if vol.dtype == torch.float32 else vol.float()
        vol = vol if vol.is_contiguous() else vol.contiguous()
        domain = domain if domain.is_contiguous() else domain.contiguous()
        nodes, _, count, seq_ps = ops.octree_build(domain, self.fixed_length, self.norm_factor)
        seq_img = ops.octree_serialize(vol, nodes, count, self.patch_size)
        return seq_img, seq_ps[..., 0], seq_ps[..., 1:], nodes, count
