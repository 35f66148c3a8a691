"""recom_amd — MI355X-native fused feature-column (embedding-column) inference path.

A from-scratch gfx950 implementation of the one hot path of AlibabaResearch/recom
(``Addons>ConcatInputs`` -> ``Addons>FeatureColumnProcess[WithSymbols]`` ->
``Addons>ConcatOutputs[NoHost]``) behind the C ABI of ``include/fcp_hip.h``.

Importing the package needs no GPU; running any op does (there is no fallback).
"""
from . import plan, synth  # noqa: F401

__all__ = ["plan", "synth", "lib", "ops"]
__version__ = "0.1.0"
