"""MI355X-native panoptic hot path (sparse-voxel U-Net + instance grouping) behind the module surface the
reference model code calls.  See DESIGN.md.  The HIP library has no CPU fallback."""
__version__ = "0.1.0"
