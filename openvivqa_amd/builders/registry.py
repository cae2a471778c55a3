"""Name -> class plugin registry with the reference's surface.

Mirrors the behaviour of /root/reference/builders/registry.py:8-90 (decorator
or direct-call registration under ``obj.__name__``, ``KeyError`` on a miss,
``AssertionError`` on a duplicate) without sharing its code.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterator, Optional, Tuple


class Registry:
    def __init__(self, name: str) -> None:
        self._name = name
        self._obj_map: Dict[str, Any] = {}

    # -- registration ------------------------------------------------------
    def _do_register(self, name: str, obj: Any) -> None:
        assert name not in self._obj_map, (
            "An object named '{}' was already registered in '{}' registry!".format(name, self._name))
        self._obj_map[name] = obj

    def register(self, obj: Optional[Any] = None) -> Any:
        """``@REG.register()`` (decorator) or ``REG.register(cls)`` (call)."""
        if obj is not None:
            self._do_register(obj.__name__, obj)
            return None

        def _decorate(target: Any) -> Any:
            self._do_register(target.__name__, target)
            return target

        return _decorate

    # -- lookup ------------------------------------------------------------
    def get(self, name: str) -> Any:
        try:
            return self._obj_map[name]
        except KeyError:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name)) from None

    def __contains__(self, name: str) -> bool:
        return name in self._obj_map

    def __iter__(self) -> Iterator[Tuple[str, Any]]:
        return iter(self._obj_map.items())

    def __len__(self) -> int:
        return len(self._obj_map)

    def __repr__(self) -> str:
        rows = "\n".join("  {:<48s} {}".format(k, v) for k, v in self._obj_map.items())
        return "Registry of {}:\n{}".format(self._name, rows)

    __str__ = __repr__
