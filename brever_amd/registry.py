"""Name -> object registries (models, criteria, metrics, batch samplers).

Behavioural contract mirrored from the reference (brever/registry.py:1-23):
registering a name twice raises ``ValueError``, looking up an unknown name
raises ``KeyError``, ``keys()`` iterates in registration order.
"""


class Registry:
    def __init__(self, tag):
        self.tag = tag
        self._items = {}

    def register(self, name):
        def decorator(obj):
            if name in self._items:
                raise ValueError(f'"{name}" already in {self.tag} registry')
            self._items[name] = obj
            return obj
        return decorator

    def get(self, name):
        try:
            return self._items[name]
        except KeyError:
            raise KeyError(f'"{name}" not in {self.tag} registry') from None

    def keys(self):
        return self._items.keys()

    def __contains__(self, name):
        return name in self._items
