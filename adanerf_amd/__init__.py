"""adanerf_amd -- MI355X-native AdaNeRF inference renderer.

Python host side over the C ABI of ``libadanerf_hip.so`` (include/adanerf_hip.h).  There is no CPU
fallback: if the HIP library is missing or no GPU is present, construction fails loudly.
"""
from .build import build_library, library_path  # noqa: F401
from .renderer import (AdaNeRFError, NeuralRenderer, Settings, PREC_BF16, PREC_FP16, PREC_FP32,  # noqa: F401
                       choose_sampling, load_library)

__all__ = ["build_library", "library_path", "load_library", "NeuralRenderer", "Settings", "AdaNeRFError",
           "PREC_BF16", "PREC_FP16", "PREC_FP32", "choose_sampling"]
