from . import census, sad_ssd, zncc  # noqa: F401  (registers the plugins)
from .matching_cost import AbstractMatchingCost, ConfigError  # noqa: F401
