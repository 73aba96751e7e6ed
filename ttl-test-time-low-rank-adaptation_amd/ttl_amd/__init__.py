"""ttl_amd — MI355X-native hot path of Test-Time Low-rank adaptation (TTL).

Importing the package is cheap and GPU-free; the HIP library is loaded on first use by
``ttl_amd._lib.load()`` and there is no CPU fallback for the hot path.
"""
from .config import VitConfig, get_config, VIT_B16, VIT_L14  # noqa: F401

__all__ = ["VitConfig", "get_config", "VIT_B16", "VIT_L14"]
