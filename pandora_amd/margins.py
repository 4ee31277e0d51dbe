"""Margins a step needs around a region of interest (reference: margins/margins.py:37-158, margins/descriptors.py): every
plugin exposes ``.margins``; PandoraMachine.check_conf accumulates them (cumulative steps add up, non-cumulative ones - the
filters - count through their maximum) into ``machine.margins`` whose ``global_margins`` a tiling caller reads."""
import operator
from dataclasses import asdict, astuple, dataclass
from functools import reduce


@dataclass(order=True, frozen=True)
class Margins:
    left: int
    up: int
    right: int
    down: int

    def __post_init__(self):
        if any(m < 0 for m in self.astuple()):
            raise ValueError(f"Margins values should be positive. Got {self.astuple()}")

    def __add__(self, other):
        return Margins(*map(operator.add, self.astuple(), other.astuple()))

    def astuple(self):
        return astuple(self)

    def asdict(self):
        return asdict(self)


def uniform(value):
    return Margins(value, value, value, value)


def max_margins(margins):
    """margins.py:146-158: element-wise maximum"""
    tuples = [m.astuple() for m in margins]
    if len(tuples) == 1:
        return Margins(*tuples[0])
    return Margins(*map(max, *tuples))


class GlobalMargins:
    """margins.py:71-143"""

    def __init__(self):
        self._cumulatives = {}
        self._non_cumulatives = {}

    def add_cumulative(self, key, value):
        if key in self._non_cumulatives:
            raise KeyError(f"{key} is already a non-cumulative margins. Cumulative margins and non-cumulative margins are exclusive.")
        if not isinstance(value, Margins):
            raise ValueError(f"MarginDict only accept values of type Margins. Got {type(value)} instead.")
        self._cumulatives[key] = value

    def add_non_cumulative(self, key, value):
        if key in self._cumulatives:
            raise KeyError(f"{key} is already a cumulative margins. Cumulative margins and non-cumulative margins are exclusive.")
        if not isinstance(value, Margins):
            raise ValueError(f"MarginDict only accept values of type Margins. Got {type(value)} instead.")
        self._non_cumulatives[key] = value

    def remove_cumulative(self, key):
        del self._cumulatives[key]

    def remove_non_cumulative(self, key):
        del self._non_cumulatives[key]

    @property
    def cumulatives(self):
        return dict(self._cumulatives)

    @property
    def non_cumulatives(self):
        return dict(self._non_cumulatives)

    @property
    def global_margins(self):
        total = reduce(operator.add, self._cumulatives.values(), Margins(0, 0, 0, 0))
        return max_margins([total, *self._non_cumulatives.values()])

    def to_dict(self):
        return {"cumulative margins": {s: m.asdict() for s, m in self._cumulatives.items()},
                "non-cumulative margins": {s: m.asdict() for s, m in self._non_cumulatives.items()},
                "global margins": self.global_margins.asdict()}

    def get(self, key):
        return self._cumulatives.get(key, self._non_cumulatives.get(key))
