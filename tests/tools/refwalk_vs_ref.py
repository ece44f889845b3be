"""fg_refwalk.h (the host's restatement of libFLAC's serial reader on damaged data) against the reference binary: the error
statuses, in order, and the frames that decode (non-silent frames of the reference).  Build container only.
usage: python tests/tools/refwalk_vs_ref.py [fuzz_first] [fuzz_count]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import libflac_ref as R
from pyflac_amd import _lib
from tests import cases

L = _lib.lib()
L.flacgpu_refwalk_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
L.flacgpu_refwalk_probe.restype = C.c_int64


def ours(data, read_size):
    buf = np.frombuffer(data, np.uint8)
    errs = np.zeros(4096, np.uint32)
    frames = np.zeros(2 * 4096, np.uint64)
    nf = C.c_uint64(0)
    ne = L.flacgpu_refwalk_probe(buf.ctypes.data, buf.size, read_size, errs.ctypes.data, errs.size, frames.ctypes.data, frames.size, C.byref(nf))
    return [int(e) for e in errs[:max(ne, 0)]], [(int(frames[2 * i]), int(frames[2 * i + 1])) for i in range(nf.value)]


def ref(data, read_size):
    pcm, frames, res = R.decode(data, read_size=read_size)
    out, pos = [], 0
    for f in frames:
        blk = pcm[pos:pos + f['blocksize']]
        pos += f['blocksize']
        out.append((f['sample_number'], f['blocksize'], bool(np.any(blk != 0))))
    return res['errors'], out


def compare(name, data, read_size, verbose=True):
    we, wf = ref(data, read_size)
    ge, gf = ours(data, read_size)
    # the reference's frames that are not silence must be exactly the frames that decode here (a decoded frame of all zeros is
    # indistinguishable from inserted silence in this view: compare on sample numbers of the non-zero ones only)
    wnz = [(a, b) for a, b, nz in wf if nz]
    gset = set(gf)
    ok = ge == we and all(x in gset for x in wnz) and len(gf) <= len(wf)
    if not ok and verbose:
        print('DIFF', name, read_size)
        print('   ref errors', we)
        print('   our errors', ge)
        missing = [x for x in wnz if x not in gset]
        extra = [x for x in gf if x not in set((a, b) for a, b, _ in wf)]
        print('   frames the reference decodes and we do not:', missing[:6], ' frames only we decode:', extra[:6], len(wf), len(gf))
    return ok


if __name__ == '__main__':
    bad = 0
    for name in sorted(cases.DAMAGE_CASES):
        for rs in (8192, 1000):
            bad += 0 if compare(name, cases.damaged_stream(name), rs) else 1
    print('damage cases:', bad, 'differ')
    if len(sys.argv) > 2:
        first, count = int(sys.argv[1]), int(sys.argv[2])
        fb = 0
        srcs = {}
        for s in ('stereo', 'mono', 'surround', '32bit'):
            with open(os.path.join(cases.GOLDEN, 'data', s + '.flac'), 'rb') as f:
                srcs[s] = f.read()
        for seed in range(first, first + count):
            r = np.random.default_rng(seed)
            name = ['stereo', 'mono', 'surround', '32bit'][int(r.integers(0, 4))]
            data = bytearray(srcs[name])
            for _ in range(int(r.integers(1, 5))):
                kind = int(r.integers(0, 4))
                p = int(r.integers(8300, len(data) - 10))
                if kind == 0:
                    data[p] ^= 1 << int(r.integers(0, 8))
                elif kind == 1:
                    del data[p:p + int(r.integers(1, 300))]
                elif kind == 2:
                    data[p:p] = r.integers(0, 256, int(r.integers(1, 600)), dtype=np.uint8).tobytes()
                else:
                    n = int(r.integers(1, 200))
                    data[p:p + n] = bytes(n)
            rs = int(r.choice([8192, 1000, 4096, 333, 65536]))
            if not compare('fuzz %d %s' % (seed, name), bytes(data), rs, verbose=fb < 8):
                fb += 1
        print('fuzz %d..%d: %d differ' % (first, first + count - 1, fb))
