"""pyflac_amd -- MI355X-native FLAC encode/decode behind pyFLAC's Python API.

Drop-in for the public names of ``pyflac/__init__.py:14-26``; the heavy lifting is in libflacgpu.so
(HIP kernels + a libFLAC-compatible C ABI, see include/flacgpu.h).  Importing the codec classes requires the
built library; there is no CPU fallback.
"""
__title__ = 'pyflac_amd'
__version__ = '0.1.0'

__all__ = [
    'StreamEncoder', 'FileEncoder', 'EncoderState', 'EncoderInitException', 'EncoderProcessException',
    'StreamDecoder', 'FileDecoder', 'OneShotDecoder', 'DecoderState', 'DecoderInitException',
    'DecoderProcessException',
]


def __getattr__(name):
    # lazy: `import pyflac_amd.synth` must work before the library is built
    if name in ('StreamEncoder', 'FileEncoder', 'EncoderState', 'EncoderInitException', 'EncoderProcessException'):
        from . import encoder
        return getattr(encoder, name)
    if name in ('StreamDecoder', 'FileDecoder', 'OneShotDecoder', 'DecoderState', 'DecoderInitException',
                'DecoderProcessException'):
        from . import decoder
        return getattr(decoder, name)
    raise AttributeError(name)
