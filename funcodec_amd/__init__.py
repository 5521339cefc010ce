"""funcodec_amd -- MI355X (gfx950) native encode/decode engine for FunCodec codec models.

The hot path (SEANet encoder -> residual vector quantiser -> SEANet decoder) runs as hand-written
HIP kernels behind the C ABI in include/funcodec_amd.h; this package is the thin PyTorch-ROCm host
that keeps the reference's Speech2Token API and checkpoint format.
"""
from .config import ArchSpec, arch_from_config, recipe_config  # noqa: F401

__all__ = ["ArchSpec", "arch_from_config", "recipe_config"]
