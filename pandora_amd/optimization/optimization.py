"""AbstractOptimization (reference: optimization/optimization.py:34-123): registry, __new__(cls, img,
**cfg) dispatch on cfg["optimization_method"], optimize_cv(cv, img_left, img_right) -> cv, and the
40-pixel uniform margin the reference uses for SGM tiling (:43, marge.py:86-101)."""
from abc import ABCMeta, abstractmethod


class AbstractOptimization:
    __metaclass__ = ABCMeta

    optimization_methods_avail = {}
    cfg = None
    margins_value = (40, 40, 40, 40)  # UniformMargins(40), optimization.py:43

    @property
    def margins(self):
        from ..margins import uniform

        return uniform(40)

    def __new__(cls, _img=None, **cfg):
        if cls is AbstractOptimization:
            if isinstance(cfg.get("optimization_method"), str):
                try:
                    return super(AbstractOptimization, cls).__new__(cls.optimization_methods_avail[cfg["optimization_method"]])
                except KeyError:
                    raise KeyError("No optimization method named {} supported".format(cfg["optimization_method"]))
            raise KeyError("No optimization method named {} supported".format(cfg.get("optimization_method")))
        return super(AbstractOptimization, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name):
        def decorator(subclass):
            cls.optimization_methods_avail[short_name] = subclass
            return subclass

        return decorator

    @abstractmethod
    def desc(self):
        """Describes the optimization method"""

    @abstractmethod
    def optimize_cv(self, cv, img_left, img_right):
        """Optimize the cost volume; returns the cost volume dataset."""
