"""vlsa_amd -- MI355X-native (gfx950) language-guided patch aggregation for VLSA.

Host side mirrors the reference's model interface (VLFAN / FeatMIL / DeepMIL / VLSA); the arithmetic lives
in hand-written HIP kernels behind the C ABI of include/vlsa_hip.h.  There is no CPU fallback.
"""
from ._native import VlsaNativeError  # noqa: F401

__all__ = ["VlsaNativeError"]
