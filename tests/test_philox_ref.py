"""The checker of the dropout kernels (tests/_philox_ref.py) against the published known-answer vectors of Philox4x32-10
(Random123's kat_vectors: counter, key -> output), so that the GPU tests compare the kernels with a pinned generator."""
import numpy as np

from tests._philox_ref import multipliers, philox4x32_10

KAT = [((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want, (ctr, [hex(int(g[0])) for g in got])


def test_multipliers_shape_values_and_rate():
    m = multipliers((257, 64), 0.5, seed=1234, step=3, site=7)
    assert m.shape == (257, 64) and set(np.unique(m)) == {0.0, 2.0}
    assert abs((m > 0).mean() - 0.5) < 0.02
    m2 = multipliers((257, 64), 0.5, seed=1234, step=4, site=7)
    assert (m != m2).mean() > 0.3
    # element numbering is the flat index: any shape with the same number of elements draws the same stream
    assert np.array_equal(multipliers((16, 3), 0.25, 5, 0, 1).reshape(-1), multipliers((6, 8), 0.25, 5, 0, 1).reshape(-1))
