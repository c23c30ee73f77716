"""Device-side pieces of the reference's data transforms (src/UCF_VIT/dataloaders/).  Only the adaptive patcher is here; file
readers, tiling and load balancing of the reference stay out of scope."""
