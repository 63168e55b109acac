"""Name -> class registry used for the backbone / ODE / ODE-solver plugin points.

Mirrors the behaviour of the reference's plugin mechanism
(reference: flowmse/util/registry.py:5-34): ``register(name)`` is a class
decorator, double registration warns and replaces, unknown names raise
``ValueError``.
"""
import warnings


class Registry:
    def __init__(self, managed_thing):
        self.managed_thing = managed_thing
        self._registry = {}

    def register(self, name):
        def deco(cls):
            if name in self._registry:
                warnings.warn(
                    f"{self.managed_thing} with name '{name}' doubly registered, "
                    "old class will be replaced.")
            self._registry[name] = cls
            return cls
        return deco

    def get_by_name(self, name):
        try:
            return self._registry[name]
        except KeyError:
            raise ValueError(f"{self.managed_thing} with name '{name}' unknown.") from None

    def get_all_names(self):
        return list(self._registry.keys())
