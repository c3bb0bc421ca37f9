"""Per-tensor memo used by the host layer (hermiticity verdicts, uploaded shape tables, tf_super(ideal) images).

An entry belongs to a tensor OBJECT: it is keyed by id(tensor), holds a weak reference to it and the `_version` it was made
at, and is dropped by the weak reference's callback when the tensor dies.  It is never keyed by `data_ptr()`: the caching
allocator hands a freed address to the next tensor of that size, with `_version` 0 again.  Writes through raw
pointers (how libc3prop fills its outputs) do not bump `_version`; callers that overwrite an operator tensor in place through
the C ABI must pass a fresh tensor or call `forget(tensor)`.
"""
from __future__ import annotations

import weakref
from typing import Any, Dict, Hashable


class TensorMemo:
    def __init__(self):
        self._d: Dict[int, tuple] = {}

    def get(self, t, key: Hashable = None, default=None) -> Any:
        ent = self._d.get(id(t))
        if ent is None or ent[0]() is not t or ent[1] != getattr(t, "_version", 0):
            return default
        return ent[2].get(key, default)

    def put(self, t, key: Hashable, value) -> None:
        k = id(t)
        ver = getattr(t, "_version", 0)
        ent = self._d.get(k)
        if ent is not None and ent[0]() is t and ent[1] == ver:
            ent[2][key] = value
            return
        try:
            ref = weakref.ref(t, lambda _r, k=k, d=self._d: d.pop(k, None))
        except TypeError:  # not weak-referenceable: not cached
            return
        self._d[k] = (ref, ver, {key: value})

    def forget(self, t) -> None:
        self._d.pop(id(t), None)

    def __len__(self):
        return len(self._d)
