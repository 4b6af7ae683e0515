"""oracle/philox.py: the numpy Philox4x32-10 against the Random123 known-answer vectors, stream slicing, and the
distributions of the derived draws."""
import numpy as np

from oracle import philox


def test_random123_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds: (counter, key) -> output
    assert [hex(v) for v in philox.philox_blocks(0, 0, 0, 1)[0]] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    # counter = ffffffff x4, key = ffffffff x2
    out = philox.philox_blocks(0xffffffffffffffff, 0xffffffff, 0xffffffffffffffff, 1, blk0=0xffffffff)[0]
    assert [hex(v) for v in out] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    # counter = pi digits 243f6a88 85a308d3 13198a2e 03707344, key = a4093822 299f31d0
    out = philox.philox_blocks(0x299f31d0a4093822, 0x85a308d3, 0x0370734413198a2e, 1, blk0=0x243f6a88)[0]
    assert [hex(v) for v in out] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_stream_slices_and_distributions():
    full = philox.uniform(7, 3, 5, 1003)
    assert np.array_equal(philox.uniform(7, 3, 5, 100, first=401), full[401:501])          # any element range of a stream
    assert full.min() >= 0 and full.max() < 1 and abs(full.mean() - 0.5) < 0.03
    assert not np.array_equal(full, philox.uniform(7, 3, 6, 1003)) and not np.array_equal(full, philox.uniform(7, 4, 5, 1003))
    z = philox.normal(1, 2, 3, 1 << 16)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and np.isfinite(z).all()
    lab = philox.labels(1, 2, 3, 10000)
    assert lab.min() == 0 and lab.max() == 9
    u = philox.dropout_u(9, 1, 0, 3, 4, 2, 2)                                              # channels-last addressing
    flat = philox.uniform(9, 1, 0, 3 * 4 * 2 * 2)
    assert u[2, 3, 1, 0] == flat[((2 * 2 + 1) * 2 + 0) * 4 + 3]
    assert np.array_equal(philox.dropout_u(9, 1, 0, 1, 4, 2, 2, first_row=2), u[2:3])
