"""Backbone plugins of the HIP sampler: only ``"ncsnpp"`` (the released FlowSE configuration) is provided."""
from .ncsnpp import NCSNpp  # noqa: F401  (registers itself)
from .shared import BackboneRegistry  # noqa: F401

__all__ = ("BackboneRegistry", "NCSNpp")
