"""ctypes mirror of libFLAC's FLAC__Frame (format.h:285-475; the layout pyFLAC's cdef states in
pyflac/builder/decoder.py:146-231) and a reader of one FLAC__Subframe into plain Python values.

Test infrastructure: what a libFLAC client reads in its write callback.  Used by the GPU tests of FLAC__Frame.subframes[]
(tests/test_gpu_api.py::TestSubframes) against libflacgpu.so, and by oracle/libflac_ref.py against the reference's binary in
the build container -- the same structs on both sides."""
import ctypes as C

import numpy as np


class _Number(C.Union):
    _fields_ = [('frame_number', C.c_uint32), ('sample_number', C.c_uint64)]


class FrameHeader(C.Structure):
    _fields_ = [('blocksize', C.c_uint32), ('sample_rate', C.c_uint32),
                ('channels', C.c_uint32), ('channel_assignment', C.c_int),
                ('bits_per_sample', C.c_uint32), ('number_type', C.c_int),
                ('number', _Number), ('crc', C.c_uint8)]


class RiceContents(C.Structure):
    _fields_ = [('parameters', C.POINTER(C.c_uint32)),
                ('raw_bits', C.POINTER(C.c_uint32)),
                ('capacity_by_order', C.c_uint32)]


class PartitionedRice(C.Structure):
    _fields_ = [('order', C.c_uint32), ('contents', C.POINTER(RiceContents))]


class _ECMData(C.Union):
    _fields_ = [('partitioned_rice', PartitionedRice)]


class EntropyCodingMethod(C.Structure):
    _fields_ = [('type', C.c_int), ('data', _ECMData)]


class SubConstant(C.Structure):
    _fields_ = [('value', C.c_int64)]


class _VerbData(C.Union):
    _fields_ = [('int32', C.POINTER(C.c_int32)), ('int64', C.POINTER(C.c_int64))]


class SubVerbatim(C.Structure):
    _fields_ = [('data', _VerbData), ('data_type', C.c_int)]


class SubFixed(C.Structure):
    _fields_ = [('entropy_coding_method', EntropyCodingMethod),
                ('order', C.c_uint32), ('warmup', C.c_int64 * 4),
                ('residual', C.POINTER(C.c_int32))]


class SubLPC(C.Structure):
    _fields_ = [('entropy_coding_method', EntropyCodingMethod),
                ('order', C.c_uint32), ('qlp_coeff_precision', C.c_uint32),
                ('quantization_level', C.c_int), ('qlp_coeff', C.c_int32 * 32),
                ('warmup', C.c_int64 * 32), ('residual', C.POINTER(C.c_int32))]


class _SubData(C.Union):
    _fields_ = [('constant', SubConstant), ('fixed', SubFixed),
                ('lpc', SubLPC), ('verbatim', SubVerbatim)]


class Subframe(C.Structure):
    _fields_ = [('type', C.c_int), ('data', _SubData), ('wasted_bits', C.c_uint32)]


class FrameFooter(C.Structure):
    _fields_ = [('crc', C.c_uint16)]


class Frame(C.Structure):
    _fields_ = [('header', FrameHeader), ('subframes', Subframe * 8),
                ('footer', FrameFooter)]


def _subframe_info(sf, blocksize):
    t = sf.type
    d = {'type': ['CONSTANT', 'VERBATIM', 'FIXED', 'LPC'][t], 'wasted': sf.wasted_bits}

    def rice(ecm, order):
        po = ecm.data.partitioned_rice.order
        cont = ecm.data.partitioned_rice.contents.contents
        d['rice_method'] = ecm.type
        d['porder'] = po
        d['rice_params'] = [cont.parameters[i] for i in range(1 << po)]
        d['residual'] = np.ctypeslib.as_array(
            C.cast(sf.data.fixed.residual if t == 2 else sf.data.lpc.residual,
                   C.POINTER(C.c_int32)), shape=(blocksize - order,)).copy()

    if t == 0:
        d['value'] = sf.data.constant.value
    elif t == 2:
        d['order'] = sf.data.fixed.order
        d['warmup'] = list(sf.data.fixed.warmup[:d['order']])
        rice(sf.data.fixed.entropy_coding_method, d['order'])
    elif t == 3:
        o = sf.data.lpc.order
        d['order'] = o
        d['precision'] = sf.data.lpc.qlp_coeff_precision
        d['shift'] = sf.data.lpc.quantization_level
        d['qlp'] = list(sf.data.lpc.qlp_coeff[:o])
        d['warmup'] = list(sf.data.lpc.warmup[:o])
        rice(sf.data.lpc.entropy_coding_method, o)
    return d
