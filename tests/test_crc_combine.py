"""The GF(2) arithmetic csrc/gn_inflate.hip uses to put a member's CRC-32 together from pieces (gi_multmodp / gi_x2nmodp / kX2n: zlib's
crc32_combine restated from its definition): the same functions in Python, on the table the kernel source holds, against zlib.crc32."""
import os
import re
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POLY = 0xEDB88320


def multmodp(a, b):
    m, p = 1 << 31, 0
    while True:
        if a & m:
            p ^= b
            if (a & (m - 1)) == 0:
                break
        m >>= 1
        b = (b >> 1) ^ POLY if b & 1 else b >> 1
    return p


def source_table():
    src = open(os.path.join(ROOT, "ganon_amd", "csrc", "gn_inflate.hip")).read()
    body = re.search(r"kX2n\[32\] = \{([^}]*)\}", src).group(1)
    return [int(x.strip().rstrip("u"), 16) for x in body.split(",") if x.strip()]


def x2nmodp(tab, n, k):
    p = 1 << 31
    while n:
        if n & 1:
            p = multmodp(tab[k & 31], p)
        n >>= 1
        k += 1
    return p


def test_the_table_in_the_kernel_source_is_x_to_the_powers_of_two():
    tab = source_table()
    assert len(tab) == 32
    p = 1 << 30  # x^1, reflected
    for k in range(32):
        assert tab[k] == p
        p = multmodp(p, p)


def test_pieces_shifted_to_the_end_xor_to_the_crc_of_the_whole():
    tab = source_table()
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, size=300_000, dtype=np.uint8).tobytes()
    for n_cuts in (0, 1, 7, 60):
        cuts = sorted({0, len(data), *map(int, rng.integers(0, len(data), size=n_cuts))})
        acc = 0
        for a, b in zip(cuts, cuts[1:]):
            acc ^= multmodp(x2nmodp(tab, len(data) - b, 3), zlib.crc32(data[a:b]))
        assert acc == zlib.crc32(data)
    # a member that began earlier: its CRC so far, moved over what follows
    head, tail = data[:123_457], data[123_457:]
    assert multmodp(x2nmodp(tab, len(tail), 3), zlib.crc32(head)) ^ zlib.crc32(tail) == zlib.crc32(data)
