"""MI355X-native MSMFormer inference hot path (hypersphere decoder, MSDeformAttn pixel decoder,
vMF mean shift) -- see DESIGN.md.  Importing the package never touches the GPU; the HIP library
is loaded on first use and its absence is an error (no CPU fallback)."""
from . import synthetic  # noqa: F401

__all__ = ["synthetic", "ops", "modeling", "meta_arch", "mean_shift", "build"]
__version__ = "0.1.0"
