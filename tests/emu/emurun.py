"""gfx950emu harness (test infrastructure): load the stand-in HIP runtime IN FRONT OF libflacgpu.so in this process, then hand out the
product's own ctypes bindings.  Import this module before anything touches pyflac_amd._lib, in a process that never imports torch
(pyflac_amd._lib imports torch only so that torch's copy of the HIP runtime is the one in the process; here a stub takes its name)."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SHIM = os.path.join(HERE, 'libamdhip64.so.7')


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def load():
    if 'torch' in sys.modules and not getattr(sys.modules['torch'], '_gfx950emu_stub', False):
        raise RuntimeError('gfx950emu: torch is already imported in this process (its HIP runtime would serve libflacgpu.so)')
    if not os.path.exists(SHIM):
        build()
    shim = ctypes.CDLL(SHIM, mode=ctypes.RTLD_GLOBAL)
    shim.gfx950emu_stats_json.restype = ctypes.c_char_p
    shim.gfx950emu_last_fault.restype = ctypes.c_char_p
    shim.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    shim.hipFree.argtypes = [ctypes.c_void_p]
    shim.gfx950emu_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    shim.gfx950emu_unregister.argtypes = [ctypes.c_void_p]
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import faketorch
    faketorch.install(shim)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from pyflac_amd import _lib
    L = _lib.lib()
    return shim, L
