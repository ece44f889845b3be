"""gfx950emu (test infrastructure): runs the probe kernels of isa_probe.hip on the emulator and prints their outputs as JSON.
usage: python tests/emu/isa_probe_run.py   (tests/test_emu_isa.py holds the expectations)"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import emurun  # noqa: E402

PROBE = os.path.join(HERE, 'libisa_probe.so')
if not os.path.exists(PROBE) or os.path.getmtime(PROBE) < os.path.getmtime(os.path.join(HERE, 'isa_probe.hip')):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', '-o', PROBE, os.path.join(HERE, 'isa_probe.hip')])
if not os.path.exists(emurun.SHIM):
    emurun.build()
shim = ctypes.CDLL(emurun.SHIM, mode=ctypes.RTLD_GLOBAL)
shim.gfx950emu_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
shim.gfx950emu_last_fault.restype = ctypes.c_char_p
P = ctypes.CDLL(PROBE)
P.probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


def inputs():
    r = np.random.default_rng(950)
    a = r.integers(0, 2**32, 64, dtype=np.uint64).astype(np.uint32)
    b = r.integers(0, 2**32, 64, dtype=np.uint64).astype(np.uint32)
    c = r.integers(0, 2**32, 64, dtype=np.uint64).astype(np.uint32)
    # a few corners
    a[:6] = [0, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0x00010000]
    b[:6] = [0, 0xFFFFFFFF, 1, 0x80000000, 0x7FFFFFFF, 31]
    c[:6] = [0, 5, 0x1F, 0x0C0D0E0F, 0x08090A0B, 0x00010203]
    f = np.concatenate([r.normal(0, 1e9, 64), r.normal(0, 3e4, 64) + 7.0, r.normal(0, 1e12, 64)])
    f[0:3] = [1.0, 3.0, 0.5]
    return np.concatenate([a, b, c]), f


def run(which, arr, out_dtype, nout):
    arr = np.ascontiguousarray(arr)
    out = np.zeros(64 * nout, out_dtype)
    shim.gfx950emu_register(arr.ctypes.data, arr.nbytes)
    shim.gfx950emu_register(out.ctypes.data, out.nbytes)
    rc = P.probe_launch(which, arr.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise SystemExit('probe %d failed: %s' % (which, shim.gfx950emu_last_fault().decode()))
    return out.reshape(64, nout)


if __name__ == '__main__':
    u, f = inputs()
    res = {'int': run(0, u, np.uint32, 16).tolist(), 'lanes': run(1, u, np.uint32, 16).tolist(),
           'f64': [[float.hex(float(x)) for x in row] for row in run(2, f, np.float64, 8)], 'lds': run(3, u, np.uint32, 8).tolist()}
    print('RESULT ' + json.dumps(res))
