"""What the emulator of tests/emu makes of the less common instruction forms the library's kernels rely on -- DPP controls with row /
bank masks and bound_ctrl, SDWA, clamp, v_perm_b32, v_bitop3_b32, v_dot2_i32_i16, v_sad_u32, the fp64 division expansion,
v_mfma_f64_4x4x4, LDS sub-word reads and atomics, cross-lane reads under a partial EXEC -- held against results derived here, in
Python, from the ISA's definitions.  (The other line of evidence for the emulator is the hardware itself: its count of the VALU
instructions an encode launch executes agrees with the MI355X's SQ_INSTS_VALU to 0.1 %)"""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))

pytestmark = pytest.mark.skipif(not (os.path.exists('/opt/rocm/bin/hipcc') and os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump')),
                                reason='the probe kernels are compiled with hipcc and read through llvm-objdump')
M = 0xFFFFFFFF


@pytest.fixture(scope='module')
def probe():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'emu', 'isa_probe_run.py')], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    # the same inputs, made here without touching the emulator
    import importlib.util
    spec = importlib.util.spec_from_file_location('isa_inputs', os.path.join(ROOT, 'tests', 'emu', 'isa_probe_run.py'))
    src = open(spec.origin).read()
    ns = {}
    exec(src[src.index('def inputs():'):src.index('def run(')], {'np': np}, ns)
    u, f = ns['inputs']()
    return r, [int(x) for x in u[:64]], [int(x) for x in u[64:128]], [int(x) for x in u[128:]], f


def _fma(x, y, z):
    """x * y + z rounded once (Fraction -> float rounds to nearest even)"""
    from fractions import Fraction
    return float(Fraction(float(x)) * Fraction(float(y)) + Fraction(float(z)))


def s32(v): return v - (1 << 32) if v & 0x80000000 else v
def s16(v): return (v & 0xFFFF) - (1 << 16) if v & 0x8000 else v & 0xFFFF
def s24(v): return (v & 0xFFFFFF) - (1 << 24) if v & 0x800000 else v & 0xFFFFFF


def bitop3(tt, a, b, c):
    r = 0
    for i in range(32):
        idx = ((a >> i & 1) << 2) | ((b >> i & 1) << 1) | (c >> i & 1)          # src0 = 0xF0, src1 = 0xCC, src2 = 0xAA
        r |= ((tt >> idx) & 1) << i
    return r


def perm(a, b, sel):
    src = (a << 32) | b
    r = 0
    for k in range(4):
        s = (sel >> (8 * k)) & 0xFF
        if s <= 7: byte = (src >> (8 * s)) & 0xFF
        elif s == 8: byte = 0xFF if b >> 15 & 1 else 0
        elif s == 9: byte = 0xFF if b >> 31 & 1 else 0
        elif s == 10: byte = 0xFF if a >> 15 & 1 else 0
        elif s == 11: byte = 0xFF if a >> 31 & 1 else 0
        elif s == 12: byte = 0
        else: byte = 0xFF
        r |= byte << (8 * k)
    return r


def test_integer_forms(probe):
    r, A, B, C, _ = probe
    for l in range(64):
        a, b, c = A[l], B[l], C[l]
        got = r['int'][l]
        want = [(abs(a - b) + c) & M, a - b if a >= b else 0, min(a + b, M), perm(a, b, c & 0x0F0F0F0F), (((a << 32) | b) >> (c & 31)) & M]
        off, wd = b & 31, c & 31
        f = (a >> off) & ((1 << wd) - 1) if wd else 0
        want.append((f - (1 << wd)) & M if wd and (f >> (wd - 1)) & 1 else f)
        want.append((s16(a) * s16(b) + s16(a >> 16) * s16(b >> 16) + c) & M)
        want.append((s24(a) * s24(b) + c) & M)
        want.append(bitop3(0x78, a, b, c))
        assert bitop3(0x78, a, b, c) == a ^ (b & c)
        want.append(bitop3(0xd2, a, b, c))
        x = a & (M >> (b & 31))
        want.append(32 - x.bit_length() if x else M)
        y = (a << (b & 31)) & M
        want.append((y & -y).bit_length() - 1 if y else M)
        want.append(((a << (b & 31)) + c) & M)
        want.append((s16(a >> 16) + b) & M)
        want.append((a * b) >> 32)
        want.append(sorted([a, b, c])[1])
        assert got == want, (l, hex(a), hex(b), hex(c), [hex(x) for x in got], [hex(x) for x in want])


def test_cross_lane_forms(probe):
    r, A, _B, _C, _ = probe
    EE = 0xEEEEEEEE
    ball = sum((A[l] & 1) << l for l in range(64))
    first2 = next(l for l in range(64) if A[l] & 2)
    for l in range(64):
        row, q = l & ~15, l & 15
        want = [A[l - 1] if q >= 1 else EE, A[l - 3] if q >= 3 else 0, A[l - 1] if l > 0 else EE,
                A[row - 1] if (l >> 4) in (1, 3) else EE, A[31] if l >= 32 else EE, A[l ^ 1],
                A[row + 15 - q] if ((l >> 2) & 3) in (0, 2) else EE, ((A[l + 2] if q + 2 <= 15 else 0) + A[l]) & M,
                A[l ^ 5], A[(l * 7 + 3) & 63], A[17], A[0], ball & M, ball >> 32,
                bin(((0xF0F0F0F0 << 32) | 0x0F0F0F0F) & ((1 << l) - 1)).count('1'), A[first2] if A[l] & 2 else EE]
        assert r['lanes'][l] == want, (l, [hex(x) for x in r['lanes'][l]], [hex(x) for x in want])


def test_fp64_forms_and_the_matrix_core_chain(probe):
    r, _A, _B, _C, f = probe
    a, b, c = f[:64], f[64:128], f[128:]
    got = [[float.fromhex(x) for x in row] for row in r['f64']]
    for l in range(64):
        w = [_fma(a[l], b[l], c[l]), a[l] / b[l], math.floor(a[l] * 1e-3), float(int(c[l] * 1e-6)),
             abs(a[l]) + max(b[l], c[l]), math.ldexp(a[l], (l % 40) - 20)]
        for k in range(6):
            if w[k] is not None:
                assert got[l][k] == w[k], (l, k, got[l][k], w[k])
        assert got[l][7] == float(int(a[l]))
    # D[blk][i][j] = C + sum_k A[blk][i][k] B[blk][k][j], fused multiply-adds in ascending k; operands in lane k * 16 + blk * 4 + x
    fma = _fma
    for blk in range(4):
        for i in range(4):
            for j in range(4):
                acc = c[i * 16 + blk * 4 + j]
                for k in range(4):
                    acc = fma(a[k * 16 + blk * 4 + i], b[k * 16 + blk * 4 + j], acc)
                assert got[i * 16 + blk * 4 + j][6] == acc, (blk, i, j)


def test_lds_forms(probe):
    r, A, B, _C, _ = probe
    words = A + B
    raw = b''.join(int(w).to_bytes(4, 'little') for w in words)
    orv = [0] * 8
    for l in range(64):
        orv[l & 7] |= 1 << (l % 32)
    sums = [sum(A[l] & 0xFF for l in range(16 * g, 16 * g + 16)) for g in range(4)]
    for l in range(64):
        h = int.from_bytes(raw[2 * (l * 3 + 1):2 * (l * 3 + 1) + 2], 'little')
        sh = int.from_bytes(raw[2 * (l + 7):2 * (l + 7) + 2], 'little', signed=True) & M
        k = l & 31
        want = [words[(l * 5 + 1) & 127], orv[l & 7], sums[l >> 4], max(B), h, sh, raw[l * 2 + 1], words[2 * k] ^ ((words[2 * k + 1] << 1) & M)]
        assert r['lds'][l] == want, (l, r['lds'][l], want)
