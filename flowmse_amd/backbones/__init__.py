from .shared import BackboneRegistry
from .ncsnpp import NCSNpp

__all__ = ["BackboneRegistry", "NCSNpp"]
