"""AbstractAggregation (reference: aggregation/aggregation.py:34-128)."""
from abc import ABCMeta, abstractmethod


class AbstractAggregation:
    __metaclass__ = ABCMeta

    aggreg_methods_avail = {}

    @property
    def margins(self):
        """NullMargins (aggregation.py:43)"""
        from ..margins import uniform

        return uniform(0)
    cfg = None

    def __new__(cls, **cfg):
        if cls is AbstractAggregation:
            if isinstance(cfg.get("aggregation_method"), str):
                try:
                    return super(AbstractAggregation, cls).__new__(cls.aggreg_methods_avail[cfg["aggregation_method"]])
                except KeyError:
                    raise KeyError("No aggregation method named {} supported".format(cfg["aggregation_method"]))
            raise KeyError("No aggregation method named {} supported".format(cfg.get("aggregation_method")))
        return super(AbstractAggregation, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name):
        def decorator(subclass):
            cls.aggreg_methods_avail[short_name] = subclass
            return subclass

        return decorator

    @abstractmethod
    def desc(self):
        """Describes the aggregation method"""

    @abstractmethod
    def cost_volume_aggregation(self, img_left, img_right, cv, **cfg):
        """Aggregate the cost volume in place."""
