// flac_dec_fast.hip -- register-resident FLAC frame decoder for gfx950 (the common shapes).
//
//   fg_dec_frames_kernel   lane = frame.  Per-lane bit reader with two words of register prefetch; Rice decode;
//                          fixed and LPC restoration unified as an FIR over a register shift history with up
//                          to 12 taps (coefficients and history in VGPRs, no LDS, no scratch).  Writes
//                          frame-planar samples (wasted bits already undone).  Frames using features outside
//                          this kernel (predictor order > 12, > 32-bit subframes) are flagged status 3 and
//                          redone by fg_decode_slow_kernel (flac_dec_kernels.hip).
//   fg_dec_crc_kernel      wave = frame: CRC-16 over the frame bytes, 64 lanes over interleaved 32-bit groups.
//   fg_dec_finish_kernel   workgroup = frame: stereo undo + interleave (or planar copy), zeros for bad frames
//                          (libFLAC delivers silence on a CRC mismatch).
//
// Reference path replaced: read_subframe_*, read_residual_partitioned_rice_, FLAC__fixed_restore_signal,
// FLAC__lpc_restore_signal, undo_channel_coding inside libFLAC.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fg_dev.h"
#include "fg_types.h"

using namespace fgdev;

#define FG_DMAXO 12

namespace {

__device__ __forceinline__ uint32_t be32(uint32_t v) { return __builtin_bswap32(v); }

// MSB-first bit reader.  Invariant between calls: avail >= 32, so peek() always returns 32 valid bits.
struct FastBR {
    const uint32_t *wp;   // next word to prefetch
    u64 acc;              // left-aligned bit window
    uint32_t avail;       // valid bits in acc
    uint32_t pre0, pre1;  // prefetched words (already big-endian swapped)
    uint32_t pos;         // bits consumed from the frame start

    __device__ __forceinline__ void init(const uint8_t *base, uint32_t start_bit)
    {
        const uintptr_t a = (uintptr_t)base + (start_bit >> 3);
        const uint32_t *w = (const uint32_t *)(a & ~(uintptr_t)3);
        const uint32_t skip = (uint32_t)(a & 3) * 8 + (start_bit & 7);
        const uint32_t w0 = be32(w[0]), w1 = be32(w[1]);
        pre0 = be32(w[2]); pre1 = be32(w[3]);
        wp = w + 4;
        acc = (((u64)w0 << 32) | w1) << skip;
        avail = 64 - skip;
        pos = start_bit;
        if (avail < 32) refill();
    }
    __device__ __forceinline__ void refill()
    {
        acc |= (u64)pre0 << (32 - avail);
        avail += 32;
        pre0 = pre1;
        pre1 = be32(*wp);
        wp++;
    }
    __device__ __forceinline__ uint32_t peek() const { return (uint32_t)(acc >> 32); }
    __device__ __forceinline__ void consume(uint32_t n)   // n <= 32
    {
        acc = (n < 32) ? (acc << n) : (acc << 31) << 1;
        avail -= n;
        pos += n;
        if (avail < 32) refill();
    }
    __device__ __forceinline__ uint32_t bits(uint32_t n)   // n <= 32
    {
        if (n == 0) return 0;
        const uint32_t v = peek() >> (32 - n);
        consume(n);
        return v;
    }
    __device__ __forceinline__ int32_t sbits(uint32_t n)
    {
        if (n == 0) return 0;
        const int32_t v = (int32_t)peek() >> (32 - n);
        consume(n);
        return v;
    }
    __device__ __forceinline__ uint32_t unary(uint32_t limit_bits)
    {
        uint32_t z = 0;
        for (;;) {
            const uint32_t p = peek();
            if (p) { const uint32_t l = (uint32_t)__clz(p); z += l; consume(l + 1); return z; }
            z += 32; consume(32);
            if (pos > limit_bits) return z;
        }
    }
};

template <bool WIDE>
__global__ void __launch_bounds__(64)
fg_dec_frames_kernel(const uint8_t *stream, const FgDecFrame *frames, uint32_t nframes, int32_t *scratch, FgDecResult *results)
{
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= nframes) return;
    const FgDecFrame fr = frames[f];
    if (fr.bytes == 0) return;                        // header already rejected
    uint32_t err = 0;
    const uint32_t n = fr.n, C = fr.channels;
    int32_t *planar = scratch + fr.out_off * C;
    if (fr.bytes < fr.hdr_bytes + 2) { results[f].err = 1; return; }
    const uint32_t end_bits = (fr.bytes - 2) * 8;
    FastBR br;
    br.init(stream + fr.byte_off, fr.hdr_bytes * 8);
    for (uint32_t ch = 0; ch < C && !err; ch++) {
        uint32_t sb = fr.bps;
        if ((fr.ca == 1 && ch == 1) || (fr.ca == 2 && ch == 0) || (fr.ca == 3 && ch == 1)) sb++;
        const uint32_t hdr = br.bits(8);
        uint32_t wasted = 0;
        if (hdr & 0x80) { err = 1; break; }
        if (hdr & 1) { wasted = br.unary(end_bits) + 1; if (wasted >= sb) { err = 1; break; } sb -= wasted; }
        if (sb > 32) { err = 3; break; }
        const uint32_t t = (hdr >> 1) & 0x3F;
        int32_t *dst = planar + (size_t)ch * n;
        // mode 0 constant, 1 verbatim, 2 predicted
        uint32_t mode, order = 0;
        if (t == 0) mode = 0;
        else if (t == 1) mode = 1;
        else if (t >= 8 && t <= 12) { mode = 2; order = t & 7; }
        else if (t >= 32) { mode = 2; order = (t & 31) + 1; }
        else { err = 1; break; }
        if (order > n) { err = 1; break; }
        if (order > FG_DMAXO) { err = 3; break; }
        if (mode == 0) {
            const int32_t v = (int32_t)((uint32_t)br.sbits(sb) << wasted);
            for (uint32_t i = 0; i < n; i++) dst[i] = v;
            continue;
        }
        if (mode == 1) {
            for (uint32_t i = 0; i < n; i++) dst[i] = (int32_t)((uint32_t)br.sbits(sb) << wasted);
            if (br.pos > end_bits) err = 1;
            continue;
        }
        // predicted: warm-up, coefficients, Rice-coded residual
        int32_t h[FG_DMAXO], q[FG_DMAXO];
#pragma unroll
        for (int j = 0; j < FG_DMAXO; j++) { h[j] = 0; q[j] = 0; }
        for (uint32_t i = 0; i < order; i++) {
            const int32_t v = br.sbits(sb);
            dst[i] = (int32_t)((uint32_t)v << wasted);
#pragma unroll
            for (int j = FG_DMAXO - 1; j > 0; j--) h[j] = h[j - 1];
            h[0] = v;
        }
        int shift = 0;
        if (t >= 32) {
            const uint32_t prec = br.bits(4) + 1;
            if (prec == 16) { err = 1; break; }
            shift = br.sbits(5);
            if (shift < 0) { err = 1; break; }
#pragma unroll
            for (int j = 0; j < FG_DMAXO; j++) if ((uint32_t)j < order) q[j] = br.sbits(prec);
        }
        else {
            // fixed predictor of order k as FIR with binomial coefficients
            const int32_t FX[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = (order == 0) ? 0 : (order == 1) ? FX[1][j] : (order == 2) ? FX[2][j] : (order == 3) ? FX[3][j] : FX[4][j];
        }
        const uint32_t method = br.bits(2);
        if (method > 1) { err = 1; break; }
        const uint32_t po = br.bits(4);
        const uint32_t plen = method ? 5 : 4, esc = method ? 31 : 15;
        const uint32_t psz = n >> po;
        if ((po > 0 && ((n & ((1u << po) - 1)) != 0 || psz < order)) || (po == 0 && n < order)) { err = 1; break; }
        uint32_t left = 0, part = 0, k = 0, raw = 0;
        bool is_esc = false;
        for (uint32_t i = order; i < n; i++) {
            while (left == 0) {
                left = (po == 0) ? (n - order) : ((part == 0) ? (psz - order) : psz);
                part++;
                k = br.bits(plen);
                is_esc = (k == esc);
                if (is_esc) raw = br.bits(5);
            }
            left--;
            int32_t r;
            if (is_esc) r = br.sbits(raw);
            else {
                const uint32_t p = br.peek();
                const uint32_t lz = p ? (uint32_t)__clz(p) : 32;
                uint32_t u;
                if (lz + 1 + k <= 32) {
                    const uint32_t rest = (lz + 1 < 32) ? (p << (lz + 1)) : 0;
                    u = (lz << k) | (k ? (rest >> (32 - k)) : 0);
                    br.consume(lz + 1 + k);
                }
                else {
                    const uint32_t msb = br.unary(end_bits);
                    u = (msb << k) | br.bits(k);
                }
                r = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
            int32_t v;
            if (!WIDE) {
                int32_t sum = 0;
#pragma unroll
                for (int j = 0; j < FG_DMAXO; j++) sum += q[j] * h[j];
                v = r + (sum >> shift);
            }
            else {
                i64 sum = 0;
#pragma unroll
                for (int j = 0; j < FG_DMAXO; j++) sum += (i64)q[j] * (i64)h[j];
                v = (int32_t)((i64)r + (sum >> shift));
            }
#pragma unroll
            for (int j = FG_DMAXO - 1; j > 0; j--) h[j] = h[j - 1];
            h[0] = v;
            dst[i] = (int32_t)((uint32_t)v << wasted);
            if (br.pos > end_bits) { err = 1; break; }
        }
    }
    if (!err) {
        const uint32_t endb = (br.pos + 7) & ~7u;
        if (endb != end_bits) err = 1;
    }
    results[f].err = err;
}

// CRC-16 (poly 0x8005, init 0) of frame bytes [0, bytes-2), compared with the stored big-endian CRC.
__global__ void __launch_bounds__(256)
fg_dec_crc_kernel(const uint8_t *stream, const FgDecFrame *frames, uint32_t nframes, FgDecResult *results, const uint16_t *crctab)
{
    __shared__ uint16_t crct[768];
    __shared__ uint32_t mult[64];
    for (int j = threadIdx.x; j < 768; j += 256) crct[j] = crctab[j];
    if (threadIdx.x < 64) mult[threadIdx.x] = crctab[768 + threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    const uint32_t fb = frames[f].bytes;
    if (fb < 3) return;
    const uint8_t *fp = stream + frames[f].byte_off;
    const uint16_t *t0 = crct, *thi = crct + 256, *tlo = crct + 512;
    const uint32_t nbytes = fb - 2;
    // process from the first 4-byte aligned address: head bytes serially, then interleaved words, then the tail
    const uint32_t mis = (uint32_t)((uintptr_t)fp & 3);
    const uint32_t head = mis ? (4 - mis) : 0;
    const uint32_t hb = head < nbytes ? head : nbytes;
    const uint32_t W = (nbytes - hb) >> 2, tail = (nbytes - hb) & 3;
    const uint32_t *wptr = (const uint32_t *)(fp + hb);
    const uint32_t pad = (64 - (W & 63)) & 63, T = (W + pad) >> 6;
    uint32_t s = 0;
    for (uint32_t t = 0; t < T; t++) {
        const int qi = (int)(t * 64 + lane) - (int)pad;
        uint32_t wv = 0;
        if (qi >= 0) wv = be32(wptr[qi]);
        s = thi[s >> 8] ^ tlo[s & 0xFF];
        uint32_t cw = 0;
        cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 24)) & 0xFF];
        cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 16)) & 0xFF];
        cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 8)) & 0xFF];
        cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ wv) & 0xFF];
        s ^= cw;
    }
    s = gf16_mul(s, mult[63 - lane]);
    uint32_t body = wave_xor32(s);
    // crc(head || body) = crc(head) * x^(32 W) + crc(body): fold the head in by running it through W zero words
    uint32_t crc = 0;
    for (uint32_t b = 0; b < hb; b++) crc = ((crc << 8) & 0xFFFF) ^ t0[((crc >> 8) ^ fp[b]) & 0xFF];
    if (hb) {
        // multiply crc by x^(32 W) with square-and-multiply over x^32 powers
        uint32_t base = mult[1], e = W, acc = 1;     // mult[1] = x^32
        bool first = true;
        while (e) {
            if (e & 1) { acc = first ? base : gf16_mul(acc, base); first = false; }
            base = gf16_mul(base, base);
            e >>= 1;
        }
        if (W) crc = gf16_mul(crc, acc);
    }
    crc ^= body;
    for (uint32_t b = 0; b < tail; b++) crc = ((crc << 8) & 0xFFFF) ^ t0[((crc >> 8) ^ fp[hb + W * 4 + b]) & 0xFF];
    const uint32_t stored = ((uint32_t)fp[nbytes] << 8) | fp[nbytes + 1];
    if (lane == 0) {
        results[f].crc = crc;
        if (results[f].err == 0 && crc != stored) results[f].err = 2;
    }
}

// Stereo undo + interleave.  One workgroup per frame.
__global__ void __launch_bounds__(256)
fg_dec_finish_kernel(const FgDecFrame *frames, uint32_t nframes, const int32_t *scratch, int32_t *out, const FgDecResult *results,
                     uint32_t interleave)
{
    const uint32_t f = blockIdx.x;
    if (f >= nframes) return;
    const FgDecFrame fr = frames[f];
    if (fr.bytes == 0 || fr.n == 0) return;
    const uint32_t status = results[f].err;
    const uint32_t n = fr.n, C = fr.channels, ca = fr.ca;
    const int32_t *pl = scratch + fr.out_off * C;
    int32_t *o = out + fr.out_off * C;
    if (C == 2) {
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            int32_t a = 0, b = 0;
            if (status == 0) {
                a = pl[i]; b = pl[n + i];
                if (ca == 1) b = a - b;
                else if (ca == 2) a = a + b;
                else if (ca == 3) {
                    const i64 side = b;
                    const i64 mid = (i64)(((u64)(i64)a) << 1) | (side & 1);
                    a = (int32_t)((mid + side) >> 1);
                    b = (int32_t)((mid - side) >> 1);
                }
            }
            if (interleave) ((int2 *)o)[i] = make_int2(a, b);
            else { o[i] = a; o[n + i] = b; }
        }
    }
    else {
        for (uint32_t j = threadIdx.x; j < n * C; j += 256) {
            const uint32_t c = j / n, i = j % n;
            const int32_t v = status == 0 ? pl[j] : 0;
            if (interleave) o[(size_t)i * C + c] = v; else o[j] = v;
        }
    }
}

}  // namespace

extern "C" int fg_launch_decode_fast(const uint8_t *d_stream, const FgDecFrame *d_frames, uint32_t nframes, int32_t *d_scratch,
                                     FgDecResult *d_results, int wide, hipStream_t stream)
{
    if (nframes == 0) return 0;
    const uint32_t nwg = (nframes + 63) / 64;
    if (wide) hipLaunchKernelGGL(fg_dec_frames_kernel<true>, dim3(nwg), dim3(64), 0, stream, d_stream, d_frames, nframes, d_scratch, d_results);
    else hipLaunchKernelGGL(fg_dec_frames_kernel<false>, dim3(nwg), dim3(64), 0, stream, d_stream, d_frames, nframes, d_scratch, d_results);
    return (int)hipGetLastError();
}

extern "C" int fg_launch_decode_finish(const uint8_t *d_stream, const FgDecFrame *d_frames, uint32_t nframes, const int32_t *d_scratch,
                                       int32_t *d_pcm, FgDecResult *d_results, const uint16_t *d_crctab, uint32_t interleave,
                                       hipStream_t stream)
{
    if (nframes == 0) return 0;
    hipLaunchKernelGGL(fg_dec_crc_kernel, dim3((nframes + 3) / 4), dim3(256), 0, stream, d_stream, d_frames, nframes, d_results, d_crctab);
    hipLaunchKernelGGL(fg_dec_finish_kernel, dim3(nframes), dim3(256), 0, stream, d_frames, nframes, d_scratch, d_pcm, d_results, interleave);
    return (int)hipGetLastError();
}
