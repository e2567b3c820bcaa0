"""String-keyed class registry (reference sgmse/util/registry.py:5-34): ``register(name)`` decorator,
``get_by_name`` raising ``ValueError`` for unknown names, ``get_all_names``."""
import warnings
from typing import Callable, Dict, List


class Registry:
    def __init__(self, managed_thing: str):
        self.managed_thing = managed_thing
        self._registry: Dict[str, type] = {}

    def register(self, name: str) -> Callable:
        def wrap(cls):
            if name in self._registry:
                warnings.warn(f"{self.managed_thing} with name '{name}' doubly registered, old class will be replaced.")
            self._registry[name] = cls
            return cls
        return wrap

    def get_by_name(self, name: str):
        if name in self._registry:
            return self._registry[name]
        raise ValueError(f"{self.managed_thing} with name '{name}' unknown.")

    def get_all_names(self) -> List[str]:
        return list(self._registry.keys())
