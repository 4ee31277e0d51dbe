"""How many pixels around a region of interest a pipeline needs (what `dist.run_row_tiled` pads its row tiles with, and the
"margins" block `pandora_amd.main` writes next to the configuration; reference behaviour: margins/margins.py, tests/test_pandora.py:150-210).

One ledger per machine: every configured step books the border it needs, either as a step whose need ADDS to the others'
(the cost / aggregation / optimisation / disparity / refinement chain: each consumes the border the previous one produced) or as
one that only has to FIT inside the total (the disparity-map filters).  The pipeline's border is then
max(sum of the adding steps, each fitting step) per side."""
from typing import NamedTuple


class _Sides(NamedTuple):
    left: int
    up: int
    right: int
    down: int


class Margins(_Sides):
    """Border widths in pixels, never negative.  Comparable and hashable as the 4-tuple (left, up, right, down)."""
    __slots__ = ()

    def __new__(cls, left, up, right, down):
        if min(left, up, right, down) < 0:
            raise ValueError(f"Margins values should be positive. Got {(left, up, right, down)}")
        return super().__new__(cls, left, up, right, down)

    def __add__(self, other):  # side by side sum (tuple concatenation makes no sense for borders)
        return Margins(*(a + b for a, b in zip(self, other)))

    def __or__(self, other):  # side by side maximum
        return Margins(*(max(a, b) for a, b in zip(self, other)))

    def astuple(self):
        return tuple(self)

    def asdict(self):
        return self._asdict()


NO_MARGIN = Margins(0, 0, 0, 0)


def uniform(value):
    return Margins(value, value, value, value)


def max_margins(margins):
    out = NO_MARGIN
    for m in margins:
        out = out | m
    return out


class GlobalMargins:
    """The ledger.  `book[step] = (margins, adds)`; a step is booked once, under one of the two kinds."""

    def __init__(self):
        self.book = {}

    def _enter(self, step, value, adds):
        if not isinstance(value, Margins):
            raise ValueError(f"a step's margins must be a Margins, got {type(value).__name__}")
        known = self.book.get(step)
        if known is not None and known[1] != adds:
            kinds = ("non-cumulative", "cumulative")
            raise KeyError(f"{step} is already booked as {kinds[known[1]]} margins; it cannot also be {kinds[adds]}")
        self.book[step] = (value, adds)

    def add_cumulative(self, step, value):
        self._enter(step, value, True)

    def add_non_cumulative(self, step, value):
        self._enter(step, value, False)

    def _drop(self, step, adds):
        if step not in self.book or self.book[step][1] != adds:
            raise KeyError(step)
        del self.book[step]

    def remove_cumulative(self, step):
        self._drop(step, True)

    def remove_non_cumulative(self, step):
        self._drop(step, False)

    def _of_kind(self, adds):
        return {step: m for step, (m, kind) in self.book.items() if kind == adds}

    cumulatives = property(lambda self: self._of_kind(True))
    non_cumulatives = property(lambda self: self._of_kind(False))

    def get(self, step):
        entry = self.book.get(step)
        return entry[0] if entry else None

    @property
    def global_margins(self):
        total, widest = NO_MARGIN, NO_MARGIN
        for m, adds in self.book.values():
            if adds:
                total = total + m
            else:
                widest = widest | m
        return total | widest

    def to_dict(self):
        """The block the reference saves in the output configuration (same keys)."""
        as_dicts = lambda kind: {step: m.asdict() for step, m in self._of_kind(kind).items()}  # noqa: E731
        return {"cumulative margins": as_dicts(True), "non-cumulative margins": as_dicts(False),
                "global margins": self.global_margins.asdict()}
