"""vector_quantize_pytorch_amd -- MI355X-native drop-in for the VectorQuantize / ResidualVQ forward
path of lucidrains/vector-quantize-pytorch.  HIP kernels: csrc/vqhip.hip, C ABI:
include/vqhip.h."""
from .codebook import Codebook
from .vector_quantize import VectorQuantize, LossBreakdown
from .residual_vq import ResidualVQ, GroupedResidualVQ
from .sim_vq import SimVQ, ResidualSimVQ
from .callers import HierarchicalVQ, RandomProjectionQuantizer

__all__ = ["VectorQuantize", "ResidualVQ", "GroupedResidualVQ", "SimVQ", "ResidualSimVQ", "RandomProjectionQuantizer", "HierarchicalVQ", "Codebook", "LossBreakdown"]
