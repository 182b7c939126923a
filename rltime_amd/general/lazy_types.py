"""Lazy "name -> module:attribute" tables behind the plugin registry: a group
is declared as a dict of import paths and resolved on first use, so importing a
registry group never drags in torch-heavy modules it does not need."""
import importlib


class LazyTypes(dict):
    def __init__(self, table):
        super().__init__()
        self._table = dict(table)

    def _load(self, name):
        module, attr = self._table[name].split(":")
        return getattr(importlib.import_module(module), attr)

    def __contains__(self, name):
        return name in self._table

    def __getitem__(self, name):
        if not dict.__contains__(self, name):
            dict.__setitem__(self, name, self._load(name))
        return dict.__getitem__(self, name)

    def keys(self):
        return self._table.keys()

    def __iter__(self):
        return iter(self._table)

    def __len__(self):
        return len(self._table)
