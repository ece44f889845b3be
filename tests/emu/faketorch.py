"""gfx950emu harness (test infrastructure): a stand-in for the few torch names the product's batch front end and the GPU tests use,
so that they run UNMODIFIED in a process whose "device" is the interpreter of tests/emu.  A tensor is a numpy array; a "cuda" tensor
is a numpy array over memory from the stand-in runtime's hipMalloc (which the interpreter's address checks know).  Installed as
`torch` in sys.modules by tests/emu/emurun.py before pyflac_amd is imported; never importable as torch anywhere else."""
import ctypes
import types

import numpy as np

_gfx950emu_stub = True
_shim = None

int8, uint8, int16, int32, int64, uint16, uint32, uint64, float32, float64 = (np.dtype(n) for n in
    ('int8', 'uint8', 'int16', 'int32', 'int64', 'uint16', 'uint32', 'uint64', 'float32', 'float64'))
float = float32
double = float64
long = int64
int = int32
short = int16


class device:
    def __init__(self, kind='cpu', index=None):
        if isinstance(kind, device):
            kind, index = kind.type, kind.index
        kind = str(kind)
        if ':' in kind:
            kind, idx = kind.split(':')
            index = builtins_int(idx)
        self.type = kind
        self.index = 0 if (kind == 'cuda' and index is None) else index

    def __eq__(self, other):
        other = device(other) if not isinstance(other, device) else other
        return self.type == other.type and (self.index or 0) == (other.index or 0)

    def __hash__(self):
        return hash((self.type, self.index or 0))

    def __repr__(self):
        return "device(type='%s'%s)" % (self.type, '' if self.index is None else ', index=%d' % self.index)


import builtins as _b
builtins_int = _b.int


class _DevMem:
    """memory from the stand-in runtime's hipMalloc; freed with the last array that views it"""
    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        if _shim.hipMalloc(ctypes.byref(p), max(nbytes, 1)) != 0:
            raise MemoryError('gfx950emu hipMalloc(%d)' % nbytes)
        self.ptr = p.value
        self.buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(self.ptr)

    def __del__(self):
        try:
            _shim.hipFree(ctypes.c_void_p(self.ptr))
        except Exception:
            pass


def _dev_array(shape, dtype):
    dtype = np.dtype(dtype)
    n = builtins_int(np.prod(shape)) if np.ndim(shape) or isinstance(shape, (tuple, list)) else builtins_int(shape)
    mem = _DevMem(n * dtype.itemsize)
    a = np.frombuffer(mem.buf, dtype=dtype, count=n).reshape(shape)
    return a, mem


class Tensor:
    def __init__(self, a, cuda=False, mem=None):
        self._a = a
        self.is_cuda = cuda
        self._mem = mem          # keeps the device allocation alive for every view

    # ---- facts
    @property
    def dtype(self): return self._a.dtype
    @property
    def shape(self): return tuple(self._a.shape)
    @property
    def device(self): return device('cuda', 0) if self.is_cuda else device('cpu')
    def dim(self): return self._a.ndim
    def numel(self): return builtins_int(self._a.size)
    def size(self, d=None): return self.shape if d is None else self.shape[d]
    def element_size(self): return self._a.dtype.itemsize
    def is_contiguous(self): return bool(self._a.flags['C_CONTIGUOUS'])
    def data_ptr(self): return builtins_int(self._a.ctypes.data)
    def __len__(self): return self._a.shape[0]

    # ---- movement
    def cuda(self, *a, **k):
        if self.is_cuda:
            return self
        arr, mem = _dev_array(self._a.shape, self._a.dtype)
        arr[...] = self._a
        return Tensor(arr, True, mem)

    def cpu(self):
        if not self.is_cuda:
            return self
        synchronize()
        return Tensor(np.array(self._a, copy=True), False)

    def to(self, *args, **kw):
        t = self
        for x in list(args) + list(kw.values()):
            if isinstance(x, np.dtype):
                t = Tensor(t._a.astype(x), t.is_cuda, None) if not t.is_cuda else t._astype_dev(x)
            elif isinstance(x, (device, str)):
                t = t.cuda() if device(x).type == 'cuda' else t.cpu()
        return t

    def _astype_dev(self, dt):
        synchronize()
        arr, mem = _dev_array(self._a.shape, dt)
        arr[...] = self._a.astype(dt)
        return Tensor(arr, True, mem)

    def numpy(self):
        if self.is_cuda:
            raise TypeError("can't convert cuda tensor to numpy; use .cpu() first")
        return self._a

    def clone(self):
        synchronize()
        if self.is_cuda:
            arr, mem = _dev_array(self._a.shape, self._a.dtype)
            arr[...] = self._a
            return Tensor(arr, True, mem)
        return Tensor(np.array(self._a, copy=True), False)

    def contiguous(self):
        return self if self.is_contiguous() else self.clone()

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return Tensor(self._a.reshape(shape), self.is_cuda, self._mem)

    view = reshape

    def flatten(self): return self.reshape(-1)

    def repeat(self, *reps):
        r = Tensor(np.tile(self._host(), reps), False)
        return r.cuda() if self.is_cuda else r

    # ---- values (the device is synchronous from the host's point of view once synchronised)
    def _host(self):
        if self.is_cuda:
            synchronize()
        return self._a

    def item(self): return self._host().item()
    def tolist(self): return self._host().tolist()
    def max(self): return Tensor(np.asarray(self._host().max()), False)
    def min(self): return Tensor(np.asarray(self._host().min()), False)
    def sum(self): return Tensor(np.asarray(self._host().sum()), False)
    def any(self): return Tensor(np.asarray(self._host().any()), False)
    def all(self): return Tensor(np.asarray(self._host().all()), False)
    def abs(self): return Tensor(np.abs(self._host()), False)
    def __int__(self): return builtins_int(self._host())
    def __index__(self): return builtins_int(self._host())
    def __bool__(self): return bool(self._host())
    def __float__(self): return _b.float(self._host())

    def __getitem__(self, k):
        if isinstance(k, Tensor):
            k = k._host()
        elif isinstance(k, tuple):
            k = tuple(x._host() if isinstance(x, Tensor) else x for x in k)
        r = self._a[k]
        if isinstance(r, np.ndarray) and r.base is not None or (isinstance(r, np.ndarray) and r.ndim):
            return Tensor(r, self.is_cuda, self._mem)
        return Tensor(np.asarray(r), False)

    def __setitem__(self, k, v):
        if self.is_cuda:
            synchronize()
        self._a[k] = v._host() if isinstance(v, Tensor) else v

    def zero_(self):
        self._host()[...] = 0
        return self

    def fill_(self, v):
        self._host()[...] = v
        return self

    def copy_(self, other):
        self._host()[...] = other._host()
        return self

    def _bin(self, other, fn):
        o = other._host() if isinstance(other, Tensor) else other
        return Tensor(np.asarray(fn(self._host(), o)), False)

    def __eq__(self, o): return self._bin(o, lambda a, b: a == b)
    def __ne__(self, o): return self._bin(o, lambda a, b: a != b)
    def __lt__(self, o): return self._bin(o, lambda a, b: a < b)
    def __le__(self, o): return self._bin(o, lambda a, b: a <= b)
    def __gt__(self, o): return self._bin(o, lambda a, b: a > b)
    def __ge__(self, o): return self._bin(o, lambda a, b: a >= b)
    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __and__(self, o): return self._bin(o, lambda a, b: a & b)
    def __or__(self, o): return self._bin(o, lambda a, b: a | b)
    def __xor__(self, o): return self._bin(o, lambda a, b: a ^ b)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __lshift__(self, o): return self._bin(o, lambda a, b: a << b)
    def __neg__(self): return Tensor(np.asarray(-self._host()), False)
    def __invert__(self): return Tensor(np.asarray(~self._host()), False)
    __radd__ = __add__
    __rmul__ = __mul__
    def __rshift__(self, o): return self._bin(o, lambda a, b: a >> b)
    __hash__ = object.__hash__

    def __repr__(self):
        return 'faketensor(%r, cuda=%s)' % (self._a, self.is_cuda)


def from_numpy(a):
    return Tensor(a, False)


def frombuffer(buf, dtype=uint8, count=-1, offset=0):
    return Tensor(np.frombuffer(buf, dtype=dtype, count=count, offset=offset), False)


def tensor(data, dtype=None, device=None):
    t = Tensor(np.array(data, dtype=dtype), False)
    return t.cuda() if device is not None and globals()['device'](device).type == 'cuda' else t


def _new(shape, dtype, dev, fill):
    if isinstance(shape, builtins_int):
        shape = (shape,)
    dtype = np.dtype(dtype if dtype is not None else float32)
    if dev is not None and device(dev).type == 'cuda':
        arr, mem = _dev_array(tuple(shape), dtype)
        if fill is not None:
            arr[...] = fill
        return Tensor(arr, True, mem)
    return Tensor(np.zeros(shape, dtype) if fill is not None else np.empty(shape, dtype), False)


def empty(*shape, dtype=None, device=None):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
        shape = tuple(shape[0])
    return _new(tuple(shape), dtype, device, None)


def zeros(*shape, dtype=None, device=None):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
        shape = tuple(shape[0])
    return _new(tuple(shape), dtype, device, 0)


def zeros_like(t):
    return _new(t.shape, t.dtype, t.device, 0)


def empty_like(t):
    return _new(t.shape, t.dtype, t.device, None)


def equal(a, b):
    return a.shape == b.shape and bool(np.array_equal(a._host(), b._host()))


def cat(ts, dim=0):
    anyc = any(t.is_cuda for t in ts)
    r = Tensor(np.concatenate([t._host() for t in ts], axis=dim), False)
    return r.cuda() if anyc else r


def synchronize(*a, **k):
    if _shim is not None:
        rc = _shim.hipDeviceSynchronize()
        if rc != 0:
            msg = _shim.gfx950emu_last_fault().decode()
            _shim.gfx950emu_clear_fault()           # (reported once: the tests that follow start clean)
            raise RuntimeError('gfx950emu: device fault: %s' % msg)


class _Stream:
    def synchronize(self):
        synchronize()


class _DeviceCtx:
    def __init__(self, *a): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False


cuda = types.ModuleType('torch.cuda')
cuda.is_available = lambda: True
cuda.device_count = lambda: 1
cuda.current_stream = lambda *a, **k: _Stream()
cuda.synchronize = synchronize
cuda.set_device = lambda *a, **k: None
cuda.device = _DeviceCtx
cuda.get_device_name = lambda *a, **k: 'gfx950emu (ISA-level emulation of gfx950, no GPU)'
cuda.empty_cache = lambda: None


# (pyflac_amd/shard.py imports torch.distributed at the top; a single emulated process never initialises it)
distributed = types.ModuleType('torch.distributed')
distributed.is_available = lambda: False
distributed.is_initialized = lambda: False
distributed.get_rank = lambda *a, **k: 0
distributed.get_world_size = lambda *a, **k: 1


def install(shim):
    """Make this module `torch` for the process."""
    import sys
    global _shim
    _shim = shim
    mod = sys.modules[__name__]
    sys.modules['torch'] = mod
    sys.modules['torch.cuda'] = cuda
    sys.modules['torch.distributed'] = distributed
    return mod
