"""Batch (many blocks / many streams per launch) front end of libflacgpu, with PCM and output in HBM.

This is the extension SURVEY.md section 8b asks for beyond the libFLAC ABI: pyFLAC's callback API hands one
stream at a time to ``FLAC__stream_encoder_process_interleaved`` (``pyflac/encoder.py:115``); the batch entry
points encode / decode any number of independent streams in one launch.  torch is used only to own device
memory; the library itself takes raw device addresses.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


class FlacGpuError(RuntimeError):
    pass


def settings(level=5, channels=2, bits_per_sample=16, sample_rate=48000, blocksize=0, streamable_subset=True):
    s = _lib.Settings()
    rc = _lib.lib().flacgpu_settings_from_level(C.byref(s), level, channels, bits_per_sample, sample_rate, blocksize,
                                                1 if streamable_subset else 0)
    if rc != 0:
        raise FlacGpuError(_lib.string_table('FLAC__StreamEncoderInitStatusString', 14)[rc].decode())
    return s


class Context:
    """One device context (stream, scratch buffers, tables).  One per process per GPU."""

    def __init__(self, device=0, testhooks=False):
        # (testhooks: the context lives in the test-hooks build of the library, which reads the kernel selectors of
        # csrc/fg_types.h fg_sel() from the environment -- the cross-check tests)
        L = self._L = _lib.testhooks_lib() if testhooks else _lib.lib()
        if not torch.cuda.is_available():
            raise FlacGpuError('no GPU visible: pyflac_amd has no CPU fallback')
        self.device = device
        self._h = L.flacgpu_ctx_create(device)
        if not self._h:
            raise FlacGpuError(self._err())

    def _err(self):
        return (self._L.flacgpu_last_error() or b'').decode()

    def close(self):
        if self._h:
            self._L.flacgpu_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- encode
    def encode(self, s, pcm, stream_lengths=None, out=None, offsets=None, debug=False):
        """Encode device tensor ``pcm`` ([total_samples, channels] int32 or int16).

        ``stream_lengths``: samples per stream (streams are laid back to back in ``pcm``); default one stream.
        Returns ``(out_bytes_tensor[:total], frame_offsets_tensor[nblocks+1], EncodeStats)``.
        """
        L = self._L
        assert pcm.is_cuda and pcm.is_contiguous()
        is16 = pcm.dtype == torch.int16
        assert is16 or pcm.dtype == torch.int32
        total = pcm.shape[0]
        if stream_lengths is None:
            stream_lengths = [total]
        # (the descriptor array and the output bound of the previous call are kept: repeated calls on one layout skip them)
        key = (tuple(int(n) for n in stream_lengths), bytes(s))
        if getattr(self, '_enc_key', None) == key:
            descs, bound, nb = self._enc_cache
        else:
            descs = (_lib.StreamDesc * len(stream_lengths))()
            pos = 0
            for i, n in enumerate(stream_lengths):
                descs[i].pcm_offset = pos
                descs[i].nsamples = int(n)
                descs[i].first_frame = 0
                pos += int(n)
            assert pos == total
            nb = C.c_uint32(0)
            bound = L.flacgpu_encode_bound(C.byref(s), descs, len(stream_lengths), C.byref(nb))
            self._enc_key, self._enc_cache = key, (descs, bound, nb)
        assert sum(key[0]) == total
        if out is None or out.numel() < bound:
            out = torch.empty(max(int(bound), 1), dtype=torch.uint8, device=pcm.device)
        if offsets is None or offsets.numel() < nb.value + 1:
            offsets = torch.empty(nb.value + 1, dtype=torch.int64, device=pcm.device)
        L.flacgpu_set_debug(self._h, 1 if debug else 0)
        st = _lib.EncodeStats()
        torch.cuda.current_stream(pcm.device).synchronize()      # the library works on its own HIP streams: the input must be complete
        rc = L.flacgpu_encode_streams(self._h, C.byref(s), pcm.data_ptr(), 1 if is16 else 0, descs, len(stream_lengths),
                                      out.data_ptr(), out.numel(), offsets.data_ptr(), C.byref(st))
        if rc != 0:
            raise FlacGpuError(self._err())
        if st.error_flags:
            raise FlacGpuError('encode error flags 0x%x' % st.error_flags)
        return out, offsets[:st.nblocks + 1], st

    def md5_streams(self, pcm, bits_per_sample, stream_lengths=None):
        """STREAMINFO's md5sum of every stream of device tensor ``pcm`` ([total_samples, channels] int32 or int16): MD5 over the
        samples as libFLAC hashes them (little-endian, (bits + 7) // 8 bytes each, interleaved).  One GPU thread per stream --
        the hash is a serial chain -- on a stream of its own.  Returns ``(list of 16-byte digests, kernel milliseconds)``."""
        L = self._L
        assert pcm.is_cuda and pcm.is_contiguous() and pcm.dtype in (torch.int16, torch.int32)
        ch = 1 if pcm.dim() == 1 else pcm.shape[1]
        if stream_lengths is None:
            stream_lengths = [pcm.shape[0]]
        descs = (_lib.StreamDesc * len(stream_lengths))()
        pos = 0
        for i, n in enumerate(stream_lengths):
            descs[i].pcm_offset, descs[i].nsamples = pos, int(n)
            pos += int(n)
        assert pos == pcm.shape[0]
        out = torch.empty(16 * len(stream_lengths), dtype=torch.uint8, device=pcm.device)
        ms = C.c_float(0)
        torch.cuda.current_stream(pcm.device).synchronize()
        if L.flacgpu_md5_streams(self._h, pcm.data_ptr(), 1 if pcm.dtype == torch.int16 else 0, ch, bits_per_sample, descs, len(stream_lengths),
                                 out.data_ptr(), C.byref(ms)) != 0:
            raise FlacGpuError(self._err())
        h = out.cpu().numpy().tobytes()
        return [h[16 * i:16 * i + 16] for i in range(len(stream_lengths))], float(ms.value)

    def debug_records(self, first, n):
        from .debug import DebugRec
        buf = (DebugRec * n)()
        if self._L.flacgpu_copy_debug(self._h, buf, first, n) != 0:
            raise FlacGpuError('no debug records')
        return buf

    # -- decode
    def decode(self, stream, frame_offsets, channels, bits_per_sample, max_samples, out=None):
        """Decode frames of a device-resident byte tensor.  ``frame_offsets``: host int64 array (nframes+1), or a
        device int64 tensor (the index stays in HBM).

        Returns ``(pcm[total_samples, channels] int32 device tensor, status uint32[nframes, 2], DecodeStats)``.
        """
        L = self._L
        assert stream.is_cuda and stream.dtype == torch.uint8
        on_dev = isinstance(frame_offsets, torch.Tensor) and frame_offsets.is_cuda
        if on_dev:
            # the index stays in HBM (e.g. the offsets tensor encode() returned)
            assert frame_offsets.dtype in (torch.int64, torch.uint64) and frame_offsets.is_contiguous()
            nframes = frame_offsets.numel() - 1
        else:
            offs = np.ascontiguousarray(np.asarray(frame_offsets, dtype=np.uint64))
            nframes = offs.size - 1
        if out is None or out.numel() < max_samples * channels:
            out = torch.empty((max(int(max_samples), 1), channels), dtype=torch.int32, device=stream.device)
        status = np.zeros((max(nframes, 1), 2), np.uint32)
        st = _lib.DecodeStats()
        torch.cuda.current_stream(stream.device).synchronize()
        fn = L.flacgpu_decode_frames_dev if on_dev else L.flacgpu_decode_frames
        rc = fn(self._h, stream.data_ptr(), stream.numel(), frame_offsets.data_ptr() if on_dev else offs.ctypes.data, nframes,
                channels, bits_per_sample, out.data_ptr(), max_samples, status.ctypes.data, C.byref(st))
        if rc != 0:
            raise FlacGpuError(self._err())
        return out[:st.total_samples], status[:nframes], st


    def decode_stream(self, stream, channels, bits_per_sample, max_samples, nframes=0, first_frame_number=0, out=None, offsets_out=None):
        """Decode the audio frames of one fixed-block-size stream from its bytes alone (device uint8 tensor): the frame
        index is made on the GPU.  ``nframes``: frames the stream holds when known (STREAMINFO), 0 = count them.

        Returns ``(pcm[total_samples, channels] int32 device tensor, status uint32[nframes, 2], DecodeStats)``.
        """
        L = self._L
        assert stream.is_cuda and stream.dtype == torch.uint8 and stream.is_contiguous()
        torch.cuda.current_stream(stream.device).synchronize()      # the library works on its own HIP streams
        if out is None or out.numel() < max_samples * channels:
            out = torch.empty((max(int(max_samples), 1), channels), dtype=torch.int32, device=stream.device)
        # the status rows: one per frame when the count is known; otherwise whatever fits -- the library never writes more
        # rows than it is told there is room for, and says how many frames it found (a second call fetches the rest)
        cap = int(nframes) if nframes else max(stream.numel() // 16 + 16, 64)
        status = np.zeros((max(cap, 1), 2), np.uint32)
        st = _lib.DecodeStats()
        rc = L.flacgpu_decode_stream_dev(self._h, stream.data_ptr(), stream.numel(), int(nframes), int(first_frame_number), channels,
                                         bits_per_sample, out.data_ptr(), max_samples, status.ctypes.data, status.shape[0],
                                         offsets_out.data_ptr() if offsets_out is not None else None, C.byref(st))
        if rc != 0:
            raise FlacGpuError(self._err())
        if st.nframes > status.shape[0]:
            # (tiny frames: more of them than the guess) -- decode again with the count now known
            return self.decode_stream(stream, channels, bits_per_sample, max_samples, nframes=st.nframes,
                                      first_frame_number=first_frame_number, out=out, offsets_out=offsets_out)
        return out[:st.total_samples], status[:st.nframes], st

    def decode_streams(self, data, ranges, channels, bits_per_sample, max_samples, out=None, offsets_out=None):
        """Decode several fixed-block-size streams laid back to back in the device uint8 tensor ``data`` from their bytes alone,
        in one launch.  ``ranges``: one ``(byte_length, nframes)`` or ``(byte_length, nframes, first_frame_number)`` per stream,
        in order.  The PCM of the streams comes out back to back.

        Returns ``(pcm[total_samples, channels] int32 device tensor, status uint32[nframes, 2], DecodeStats)``.
        """
        L = self._L
        assert data.is_cuda and data.dtype == torch.uint8 and data.is_contiguous()
        torch.cuda.current_stream(data.device).synchronize()
        if out is None or out.numel() < max_samples * channels:
            out = torch.empty((max(int(max_samples), 1), channels), dtype=torch.int32, device=data.device)
        rg = (_lib.StreamRange * len(ranges))()
        pos = total = 0
        for i, r in enumerate(ranges):
            rg[i].byte_offset = pos
            rg[i].byte_length = int(r[0])
            rg[i].nframes = int(r[1])
            rg[i].first_frame_number = int(r[2]) if len(r) > 2 else 0
            pos += int(r[0])
            total += int(r[1])
        assert pos == data.numel()
        status = np.zeros((max(total, 1), 2), np.uint32)
        st = _lib.DecodeStats()
        rc = L.flacgpu_decode_streams_dev(self._h, data.data_ptr(), data.numel(), rg, len(ranges), channels, bits_per_sample,
                                          out.data_ptr(), max_samples, status.ctypes.data, status.shape[0],
                                          offsets_out.data_ptr() if offsets_out is not None else None, C.byref(st))
        if rc != 0:
            raise FlacGpuError(self._err())
        return out[:st.total_samples], status[:st.nframes], st


class MultiContext:
    """Single-process front end over several GPUs (BASELINE config 5: a batch of independent streams): stream s is encoded
    on device s mod ndevices (pyflac_amd.shard.streams_for_rank), each device encodes its share in ONE launch, and the
    frames come back in stream order.  No data-path collective: streams are independent.  With one process per GPU
    (torch.distributed, bench.py --gpus N) the same mapping is used across ranks."""

    def __init__(self, devices=None):
        if devices is None:
            devices = list(range(torch.cuda.device_count()))
        if not devices:
            raise FlacGpuError('no GPU visible: pyflac_amd has no CPU fallback')
        self.devices = list(devices)
        self.contexts = [Context(d) for d in self.devices]

    def close(self):
        for c in self.contexts:
            c.close()

    def encode_streams(self, s, streams):
        """``streams``: list of host arrays ([samples, channels] int16 / int32), one per stream.  Returns a list (stream
        order) of ``(frames: bytes, frame_sizes: list[int])``."""
        from . import shard

        def make(ctx, dev):
            def run(mine):
                if not mine:
                    return []
                with torch.cuda.device(dev):
                    host = np.concatenate([np.ascontiguousarray(x) for x in mine])
                    pcm = torch.from_numpy(host).to('cuda:%d' % dev)
                    out, offs, st = ctx.encode(s, pcm, stream_lengths=[len(x) for x in mine])
                    h_out = out[:st.total_bytes].cpu().numpy().tobytes()
                    h_offs = offs.cpu().numpy()
                res, b = [], 0
                for x in mine:
                    nb = -(-len(x) // s.blocksize)
                    lo, hi = int(h_offs[b]), int(h_offs[b + nb])
                    res.append((h_out[lo:hi], [int(v) for v in np.diff(h_offs[b:b + nb + 1])]))
                    b += nb
                return res
            return run
        return shard.encode_sharded(list(streams), [make(c, d) for c, d in zip(self.contexts, self.devices)])

    def decode_streams(self, streams, channels, bits_per_sample, blocksize):
        """``streams``: list (stream order) of ``(frames: bytes, nframes, nsamples)`` -- the audio frames of a fixed-block-size
        stream, how many there are and how many samples they hold (STREAMINFO).  Stream s is decoded on device s mod ndevices, each
        device decodes its share from the bytes alone in ONE launch.  Returns the PCM (host int32 arrays) in stream order."""
        from . import shard

        def make(ctx, dev):
            def run(mine):
                if not mine:
                    return []
                with torch.cuda.device(dev):
                    blob = np.frombuffer(b''.join(m[0] for m in mine), np.uint8).copy()
                    data = torch.from_numpy(blob).to('cuda:%d' % dev)
                    total = sum(int(m[2]) for m in mine)
                    pcm, status, st = ctx.decode_streams(data, [(len(m[0]), int(m[1])) for m in mine], channels, bits_per_sample, total)
                    if st.error_frames:
                        raise FlacGpuError('%d frames failed' % st.error_frames)
                    host = pcm.cpu().numpy()
                res, at = [], 0
                for m in mine:
                    res.append(host[at:at + int(m[2])])
                    at += int(m[2])
                return res
            return run
        return shard.run_sharded(list(streams), [make(c, d) for c, d in zip(self.contexts, self.devices)])


def index_frames(data):
    """Host frame index of a complete FLAC stream (bytes).  Returns (offsets uint64[nframes+1], StreamInfo)."""
    L = _lib.lib()
    buf = np.frombuffer(data, np.uint8)
    cap = max(len(data) // 9 + 16, 64)          # (no frame is shorter than 9 bytes)
    offs = np.zeros(cap, np.uint64)
    si = _lib.StreamInfo()
    audio = C.c_uint64(0)
    n = L.flacgpu_index_frames(buf.ctypes.data, buf.size, offs.ctypes.data, cap, C.byref(si), C.byref(audio))
    if n < 0:
        raise FlacGpuError('not a FLAC stream')
    return offs[:n + 1].copy(), si
