"""Cost-volume confidence: ambiguity, risk and interval bounds on the device, std_intensity on the host."""
from .cost_volume_confidence import AbstractCostVolumeConfidence  # noqa: F401
from .ambiguity import Ambiguity  # noqa: F401
from .std_intensity import StdIntensity  # noqa: F401
from .risk import Risk  # noqa: F401
from .interval_bounds import IntervalBounds  # noqa: F401
