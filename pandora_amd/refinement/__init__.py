from .refinement import AbstractRefinement, Quadratic, Vfit  # noqa: F401
