from . import sgm  # noqa: F401
from .optimization import AbstractOptimization  # noqa: F401
