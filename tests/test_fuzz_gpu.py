"""GPU: tools/fuzz_parity.py inside the suite (one seed, 20 random shapes) -- the assembly passes on both plans against one launch
per step, the oracle on the small shapes, the 2D backward against torch autograd through the plain-torch restatement of the
reference ops, the 3D paths against each other and the oracle."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def test_fuzz_random_shapes_one_seed():
    from tools.fuzz_parity import run
    n2, nb2, n3, nb, worst = run(cases=20, seed=4, verbose=False)
    assert n2 == 20 and n3 == 20 and nb2 >= 3 and worst <= 1e-5
