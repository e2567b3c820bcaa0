from .shared import BackboneRegistry
from .ncsnpp import NCSNpp, NCSNpp_48k, NCSNpp_v2

__all__ = ["BackboneRegistry", "NCSNpp", "NCSNpp_48k", "NCSNpp_v2"]
