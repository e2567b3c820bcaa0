from .shared import BackboneRegistry
from .ncsnpp import NCSNpp, NCSNpp_48k

__all__ = ["BackboneRegistry", "NCSNpp", "NCSNpp_48k"]
