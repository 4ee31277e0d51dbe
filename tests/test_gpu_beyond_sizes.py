"""A pair larger than every BASELINE configuration through the integer path's two independent kernel families.  A module of its
own: the eight path volumes of this size are 140 GB, and a context that ran the float32 route at the same size before (its
accumulator volume and hand-off buffer stay with the context) would not leave the room."""
import pytest

pytestmark = pytest.mark.gpu


def test_beyond_the_baseline_sizes_family_form_against_eight_volumes():
    """16384 x 16384, d = [0, 64]: 1.74e10 cells, byte volumes of 17.4 GB, cell indices far beyond 2^32 - three direction-family
    volumes against eight path volumes: identical maps bit for bit."""
    from bench import synthetic_pair
    from pandora_amd.engine import Engine
    from tests.test_gpu_fam8 import _family_against_eight_volumes

    eng = Engine(0)
    try:
        _family_against_eight_volumes(eng, synthetic_pair, 16384, 16384, 0, 64)
    finally:
        eng.close()
