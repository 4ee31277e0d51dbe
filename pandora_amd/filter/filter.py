"""AbstractFilter (reference: filter/filter.py:34-140): same registry mechanics and constructor protocol
(``AbstractFilter(cfg=..., image_shape=..., step=...)``)."""
from abc import ABCMeta, abstractmethod


class AbstractFilter:
    __metaclass__ = ABCMeta

    filter_methods_avail = {}

    @property
    def margins(self):
        """NullMargins (the reference's default for this step)"""
        from ..margins import uniform

        return uniform(0)

    cfg = None

    def __new__(cls, *args, cfg=None, step=1, **kwargs):
        if cls is AbstractFilter:
            method = (cfg or {}).get("filter_method")
            if isinstance(method, str):
                try:
                    return super(AbstractFilter, cls).__new__(cls.filter_methods_avail[method])
                except KeyError:
                    raise KeyError("No filter method named {} supported".format(method))
            raise KeyError("No filter method named {} supported".format(method))
        return super(AbstractFilter, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name):
        def decorator(subclass):
            cls.filter_methods_avail[short_name] = subclass
            return subclass

        return decorator

    @abstractmethod
    def desc(self):
        """Describes the filtering method"""

    @abstractmethod
    def filter_disparity(self, disp, img_left=None, img_right=None, cv=None):
        """Post-process the disparity map in place by filtering its valid pixels."""
