"""Backbone plugin registry (reference: flowmse/backbones/shared.py:10)."""
from flowmse_amd.util.registry import Registry

BackboneRegistry = Registry("Backbone")
