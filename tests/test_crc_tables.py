"""The kernels' CRC-16 tables are built on the host (fg_ctx.cpp fg_crc_tables_host, round 6: products and powers in GF(2)[x] / P,
a power of x by squaring) and copied to the device at context creation.  Held here, without a GPU, against the bit-by-bit
definition of FLAC's CRC-16 (polynomial x^16 + x^15 + x^2 + 1, initial value 0, no reflection: format.h:447 FLAC__crc16; the
frame footer, format.h:469-475), entry by entry over the layout the kernels index."""
import ctypes
import numpy as np

from pyflac_amd import _lib

WORDS = 2048 + 2 * 5632


def _crc16_bitwise(data, state=0):
    for byte in data:
        state ^= byte << 8
        for _ in range(8):
            state = ((state << 1) ^ 0x8005) & 0xFFFF if state & 0x8000 else (state << 1) & 0xFFFF
    return state


def _powers(n):
    """x^k mod P for k = 0 .. n - 1, one shift a step."""
    q = np.zeros(n, dtype=np.uint32)
    v = 1
    for k in range(n):
        q[k] = v
        v = ((v << 1) ^ 0x8005) & 0xFFFF if v & 0x8000 else (v << 1) & 0xFFFF
    return q


Q = _powers(8 * 4200 + 64)


def _times_xpow(s, bits):
    """(the polynomial s) * x^bits mod P: the sum over the set bits b of s of x^(b + bits)."""
    r = 0
    for b in range(16):
        if (s >> b) & 1:
            r ^= int(Q[b + bits])
    return r


def _tables():
    out = (ctypes.c_uint16 * WORDS)()
    L = _lib.lib()
    L.flacgpu_debug_crc_tables.argtypes = [ctypes.POINTER(ctypes.c_uint16)]
    L.flacgpu_debug_crc_tables.restype = None
    L.flacgpu_debug_crc_tables(out)
    return np.frombuffer(out, dtype=np.uint16).astype(np.uint32)


def test_the_test_own_arithmetic_is_the_bitwise_crc():
    r = np.random.default_rng(5)
    for n in (1, 2, 3, 7, 64, 300):
        msg = r.integers(0, 256, n).tolist()
        # CRC of a message = the message as a polynomial times x^16, mod P
        want = 0
        for j, byte in enumerate(msg):
            want ^= _times_xpow(byte, 16 + 8 * (n - 1 - j))
        assert want == _crc16_bitwise(msg)
    # a state carried over zero bytes is the state times x^(8 n)
    for s in (1, 0x8000, 0xBEEF):
        assert _crc16_bitwise([0] * 100, s) == _times_xpow(s, 800)


def test_byte_tables_and_slicing_tables():
    t = _tables()
    for i in range(256):
        assert t[i] == _crc16_bitwise([i])
        assert t[1024 + i] == _crc16_bitwise([i, 0])
        assert t[1280 + i] == _crc16_bitwise([i, 0, 0])
        assert t[1536 + i] == _crc16_bitwise([i, 0, 0, 0])


def test_contribution_tables_2048_bits_up_and_powers_of_x():
    t = _tables()
    for i in range(256):
        assert t[256 + i] == _times_xpow(i << 8, 2048)
        assert t[512 + i] == _times_xpow(i, 2048)
    for k in range(64):
        assert t[768 + k] == _times_xpow(1, 32 * k)
    assert not t[832:1024].any()
    # (what the two of them mean together: a state carried over 256 zero bytes)
    assert t[512 + 0x5A] ^ t[256 + 0xC3] == _crc16_bitwise([0] * 256, 0xC35A)


def test_the_direct_packing_path_sets():
    t = _tables()
    for s, nt in ((0, 256), (1, 128)):
        x = t[2048 + 5632 * s:2048 + 5632 * (s + 1)]
        for i in range(256):
            assert x[i] == _times_xpow(i << 8, 128 * nt)
            assert x[256 + i] == _times_xpow(i, 128 * nt)
            assert x[512 + i] == _crc16_bitwise([i, 0, 0, 0])
            assert x[768 + i] == _crc16_bitwise([i, 0, 0])
            assert x[1024 + i] == _crc16_bitwise([i, 0])
            assert x[1280 + i] == _crc16_bitwise([i])
        for rem in range(16):
            for th in range(nt):
                assert x[1536 + rem * nt + th] == _times_xpow(1, 128 * (nt - 1 - th) + 8 * rem), (s, rem, th)
        assert not x[1536 + 16 * nt:].any()
