"""BackboneRegistry (reference sgmse/backbones/shared.py:10)."""
from ..util.registry import Registry

BackboneRegistry = Registry("Backbone")
