from .disparity import AbstractDisparity, WinnerTakesAll  # noqa: F401
