from . import cbca  # noqa: F401
from .aggregation import AbstractAggregation  # noqa: F401
