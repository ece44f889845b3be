// flac_enc_fast_impl.h -- specialised one-block-per-wavefront FLAC frame encoder for the common shapes:
// 1 or 2 channels, block length a multiple of 64 with at least MAXO samples per lane, Rice partition order <= 6,
// max LPC order <= 12, bits-per-sample <= 24.  Everything else goes to the generic kernel in flac_enc_kernels.hip;
// both produce identical bytes (tests/test_gpu_encode.py).
//
// Same algorithm and stage order as the generic kernel; what changes is the mapping:
//   * lane = segment.  Lane i owns the n/64 consecutive samples [i*seg, (i+1)*seg) of every channel.  The analysis
//     passes (fixed-predictor error sums, FIR residual of every predictor candidate) and the two packing passes walk
//     the segment serially with the predictor history in registers, addressed statically (the loop is unrolled by the
//     history length), so a pass costs one LDS read per sample and channel instead of one per tap;
//   * a lane's |residual| total is the Rice partition sum of the finest partition order (or a power-of-two fraction
//     of it), so the partition sums need no wave reductions, only pairwise merges;
//   * packing: one pass gives every lane the exact bit length of its segment, one prefix sum gives its start, and a
//     second pass ORs the codes straight into an LDS window at that position -- no per-sample wave scans, no
//     divergent word emission; the window is written out coalesced, the CRC-16 advances with it;
//   * compile-time candidate set (L,R,M,S / L,R / mono), predictor coefficients in SGPRs (wave-uniform);
//   * the autocorrelation keeps libFLAC's order-preserving fp64 chains (lane = candidate x lag): one LDS read per step,
//     the lag-0 operand comes from a DPP row broadcast inside the FMA;
//   * LDS accesses of the single wave are ordered by issue, so stages are separated by compiler fences only;
//     wave-uniform state (bit position, decisions) is kept in plain locals so it stays in SGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fg_dev.h"
#include "fg_types.h"

#define FG_LN2 0.69314718055994530942
// LDS arrays are addressed through address_space(3) pointers: ds_* instructions, no FLAT aperture checks.
#define LDS __attribute__((address_space(3)))
#define FGI __device__ __forceinline__

#define FGS_FBW 1024      // frame-bit window, 32-bit words
#define FGS_DH 32         // autocorrelation: history doubles kept in front of each chunk
#define FGS_DK 128        // autocorrelation: chunk length                                   [wave-per-candidate kernel]
#define FGS_DSTR 168      // doubles per candidate row (history + chunk + bank skew)          [wave-per-candidate kernel]
#define FGS_CK 96         // single-wave kernel: chunk length (LDS is the occupancy limit there)
#define FGS_CSTR 136      // doubles per row: 32 history + 96 + read-ahead slack; 272 words = 16 banks of skew

using namespace fgdev;

namespace {

template <bool ACC64> struct FastTypes {
    typedef typename std::conditional<ACC64, u64, uint32_t>::type sum_t;
    // bits-per-sample <= 16 (the !ACC64 shapes): samples are staged as int16, halving the LDS footprint
    typedef typename std::conditional<ACC64, int32_t, int16_t>::type samp_t;
    // elements of skew between the rows of neighbouring lanes (rows stay 4-byte aligned; 33 / 66 words of stride: no bank conflicts)
    static constexpr uint32_t PADE = 2;
};

// candidate value from the two channel samples
template <bool MS, int C> FGI int32_t fcv(int32_t L, int32_t R)
{
    if (!MS) return C == 0 ? L : R;
    if (C == 0) return L;
    if (C == 1) return R;
    if (C == 2) return (L + R) >> 1;
    return L - R;
}

FGI uint32_t fabs32(int32_t v) { return (uint32_t)(v < 0 ? -v : v); }

// sum_j q[j] * h[(u - 1 - j) mod MAXO] with 24-bit multiplies: one v_mad_i32_i24 per tap, coefficient from an SGPR.
// (History slot of sample s is s mod MAXO; u is the slot of the sample being predicted, a compile-time constant after
// unrolling.)
template <int MAXO> FGI int32_t fir24(const int32_t (&q)[MAXO], const int32_t (&h)[MAXO], int u)
{
    // one asm statement for the whole sum: the hazard recogniser pads every inline-asm statement with s_nop
    int32_t sm = 0;
#define FG_H(j) h[(u - 1 - (j) + 2 * MAXO) % MAXO]
    if (MAXO == 8) {
        asm("v_mad_i32_i24 %0, %1, %9, %0\n\tv_mad_i32_i24 %0, %2, %10, %0\n\tv_mad_i32_i24 %0, %3, %11, %0\n\t"
            "v_mad_i32_i24 %0, %4, %12, %0\n\tv_mad_i32_i24 %0, %5, %13, %0\n\tv_mad_i32_i24 %0, %6, %14, %0\n\t"
            "v_mad_i32_i24 %0, %7, %15, %0\n\tv_mad_i32_i24 %0, %8, %16, %0"
            : "+v"(sm)
            : "v"(q[7 % MAXO]), "v"(q[6 % MAXO]), "v"(q[5 % MAXO]), "v"(q[4 % MAXO]), "v"(q[3 % MAXO]), "v"(q[2 % MAXO]), "v"(q[1 % MAXO]), "v"(q[0]),
              "v"(FG_H(7)), "v"(FG_H(6)), "v"(FG_H(5)), "v"(FG_H(4)), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
    }
    else {
        asm("v_mad_i32_i24 %0, %1, %7, %0\n\tv_mad_i32_i24 %0, %2, %8, %0\n\tv_mad_i32_i24 %0, %3, %9, %0\n\t"
            "v_mad_i32_i24 %0, %4, %10, %0\n\tv_mad_i32_i24 %0, %5, %11, %0\n\tv_mad_i32_i24 %0, %6, %12, %0"
            : "+v"(sm)
            : "v"(q[11 % MAXO]), "v"(q[10 % MAXO]), "v"(q[9 % MAXO]), "v"(q[8 % MAXO]), "v"(q[7 % MAXO]), "v"(q[6 % MAXO]),
              "v"(FG_H(11)), "v"(FG_H(10)), "v"(FG_H(9)), "v"(FG_H(8)), "v"(FG_H(7)), "v"(FG_H(6)));
        asm("v_mad_i32_i24 %0, %1, %7, %0\n\tv_mad_i32_i24 %0, %2, %8, %0\n\tv_mad_i32_i24 %0, %3, %9, %0\n\t"
            "v_mad_i32_i24 %0, %4, %10, %0\n\tv_mad_i32_i24 %0, %5, %11, %0\n\tv_mad_i32_i24 %0, %6, %12, %0"
            : "+v"(sm)
            : "v"(q[5 % MAXO]), "v"(q[4 % MAXO]), "v"(q[3 % MAXO]), "v"(q[2 % MAXO]), "v"(q[1 % MAXO]), "v"(q[0]),
              "v"(FG_H(5)), "v"(FG_H(4)), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
    }
#undef FG_H
    return sm;
}
// 17..25-bit samples (24-bit input, side channel included): the exact 64-bit sum without 64-bit multiplies.  A history
// sample is kept as (x >> 12) in the high half and (x & 0xFFF) in the low half of one register (fpack); with coefficients
// below 2^15 in magnitude and at most 12 taps both partial sums fit int32, and sum q*x = 4096 * sum q*hi + sum q*lo.
// v_mad_i32_i16 multiplies the 16-bit halves op_sel picks and adds a 32-bit value (full rate; v_mad_i64_i32 is not).
FGI int32_t fpack(int32_t x) { return (int32_t)(((uint32_t)(x >> 12) << 16) | ((uint32_t)x & 0xFFFu)); }
template <int MAXO> FGI i64 fir48(const int32_t (&q)[MAXO], const int32_t (&h)[MAXO], int u)
{
    int32_t sl = 0, sh = 0;
#define FG_H(j) h[(u - 1 - (j) + 2 * MAXO) % MAXO]
#define FG_M6(OPS) "v_mad_i32_i16 %0, %1, %7, %0 " OPS "\n\tv_mad_i32_i16 %0, %2, %8, %0 " OPS "\n\tv_mad_i32_i16 %0, %3, %9, %0 " OPS "\n\t" \
                   "v_mad_i32_i16 %0, %4, %10, %0 " OPS "\n\tv_mad_i32_i16 %0, %5, %11, %0 " OPS "\n\tv_mad_i32_i16 %0, %6, %12, %0 " OPS
#define FG_M4(OPS) "v_mad_i32_i16 %0, %1, %5, %0 " OPS "\n\tv_mad_i32_i16 %0, %2, %6, %0 " OPS "\n\tv_mad_i32_i16 %0, %3, %7, %0 " OPS "\n\t" \
                   "v_mad_i32_i16 %0, %4, %8, %0 " OPS
    if (MAXO == 8) {
        asm(FG_M4("op_sel:[0,0,0,0]") : "+v"(sl) : "v"(q[7 % MAXO]), "v"(q[6 % MAXO]), "v"(q[5 % MAXO]), "v"(q[4 % MAXO]), "v"(FG_H(7)), "v"(FG_H(6)), "v"(FG_H(5)), "v"(FG_H(4)));
        asm(FG_M4("op_sel:[0,0,0,0]") : "+v"(sl) : "v"(q[3 % MAXO]), "v"(q[2 % MAXO]), "v"(q[1 % MAXO]), "v"(q[0]), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
        asm(FG_M4("op_sel:[0,1,0,0]") : "+v"(sh) : "v"(q[7 % MAXO]), "v"(q[6 % MAXO]), "v"(q[5 % MAXO]), "v"(q[4 % MAXO]), "v"(FG_H(7)), "v"(FG_H(6)), "v"(FG_H(5)), "v"(FG_H(4)));
        asm(FG_M4("op_sel:[0,1,0,0]") : "+v"(sh) : "v"(q[3 % MAXO]), "v"(q[2 % MAXO]), "v"(q[1 % MAXO]), "v"(q[0]), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
    }
    else {
        asm(FG_M6("op_sel:[0,0,0,0]") : "+v"(sl) : "v"(q[11 % MAXO]), "v"(q[10 % MAXO]), "v"(q[9 % MAXO]), "v"(q[8 % MAXO]), "v"(q[7 % MAXO]), "v"(q[6 % MAXO]),
            "v"(FG_H(11)), "v"(FG_H(10)), "v"(FG_H(9)), "v"(FG_H(8)), "v"(FG_H(7)), "v"(FG_H(6)));
        asm(FG_M6("op_sel:[0,0,0,0]") : "+v"(sl) : "v"(q[5 % MAXO]), "v"(q[4 % MAXO]), "v"(q[3 % MAXO]), "v"(q[2 % MAXO]), "v"(q[1 % MAXO]), "v"(q[0]),
            "v"(FG_H(5)), "v"(FG_H(4)), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
        asm(FG_M6("op_sel:[0,1,0,0]") : "+v"(sh) : "v"(q[11 % MAXO]), "v"(q[10 % MAXO]), "v"(q[9 % MAXO]), "v"(q[8 % MAXO]), "v"(q[7 % MAXO]), "v"(q[6 % MAXO]),
            "v"(FG_H(11)), "v"(FG_H(10)), "v"(FG_H(9)), "v"(FG_H(8)), "v"(FG_H(7)), "v"(FG_H(6)));
        asm(FG_M6("op_sel:[0,1,0,0]") : "+v"(sh) : "v"(q[5 % MAXO]), "v"(q[4 % MAXO]), "v"(q[3 % MAXO]), "v"(q[2 % MAXO]), "v"(q[1 % MAXO]), "v"(q[0]),
            "v"(FG_H(5)), "v"(FG_H(4)), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
    }
#undef FG_M6
#undef FG_M4
#undef FG_H
    return (i64)(((u64)(i64)sh) << 12) + (i64)sl;
}
template <int MAXO> FGI i64 fir64(const int32_t (&q)[MAXO], const int32_t (&h)[MAXO], int u)
{
    i64 sm = 0;
#pragma unroll
    for (int j = MAXO - 1; j >= 0; j--) sm += (i64)q[j] * (i64)h[(u - 1 - j + 2 * MAXO) % MAXO];
    return sm;
}

FGI double f_ebps(double e, double scale)
{
    if (e > 0.0) {
        const double bb = 0.5 * log(scale * e) / FG_LN2;
        return bb >= 0.0 ? bb : 0.0;
    }
    else if (e < 0.0) return 1e32;
    return 0.0;
}

// ------------------------------------------------------------------ frame-bit window (LDS) -> HBM slot, CRC-16 alongside
// Bits are ORed into a zeroed window of FGS_FBW words that starts at absolute word `wbase` of the frame.  flush() writes
// the complete words out (big-endian) and advances the per-lane CRC state: lane l owns the words l, l+64, l+128, ... of the
// frame, so its state is  state * x^2048 + crc(word)  whenever it meets its next word, whatever the flush boundaries.
// where the CRC tables of a kernel live: global memory for the single-wave kernel, LDS for the wave-per-candidate one
#ifndef FG_CRCT
#define FG_CRCT
#endif
struct FrameBits {
    LDS uint32_t *w;
    const FG_CRCT uint16_t *t0, *thi, *tlo, *t1, *t2, *t3;      // byte table, x^2048 multiply tables, slicing tables
    uint32_t *outw;
    uint32_t wbase, slot_words, err;
    uint32_t crc;          // per lane
};

// OR `vbits` bits (val < 2^vbits, vbits in [0, 32]) into the window at absolute bit position `pos`
FGI void fb_or(const FrameBits &b, uint32_t pos, uint32_t val, uint32_t vbits)
{
    // value left-aligned at bit `sh` of a 64-bit big-endian pair: x = val << (64 - sh - vbits)
    const uint32_t rel = pos - (b.wbase << 5);
    const uint32_t word = rel >> 5, sh = rel & 31;
    const u64 x = (u64)val << ((64 - sh - vbits) & 63);
    __hip_atomic_fetch_or(&b.w[word], (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_fetch_or(&b.w[word + 1], (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// write out the complete words below bit position `upto`
FGI void fb_flush(FrameBits &b, int lane, uint32_t upto)
{
    const uint32_t wend = upto >> 5;
    if (wend <= b.wbase) return;
    const uint32_t nfull = wend - b.wbase;
    if (wend > b.slot_words) { b.err |= FG_ERR_SLOT; }
    wave_lds_fence();
    for (uint32_t row = b.wbase & ~63u; row < wend; row += 64) {
        const uint32_t wi = row + (uint32_t)lane;
        if (wi >= b.wbase && wi < wend) {
            const uint32_t v = b.w[wi - b.wbase];
            if (wi < b.slot_words) b.outw[wi] = __builtin_bswap32(v);
            uint32_t s = b.crc;
            s = b.thi[s >> 8] ^ b.tlo[s & 0xFF];
            // crc of the word by slicing: four independent look-ups (byte followed by 3, 2, 1, 0 zero bytes)
            b.crc = s ^ b.t3[v >> 24] ^ b.t2[(v >> 16) & 0xFF] ^ b.t1[(v >> 8) & 0xFF] ^ b.t0[v & 0xFF];
        }
    }
    const uint32_t carry = b.w[nfull];
    wave_lds_fence();
    for (uint32_t j = lane; j <= nfull + 1 && j < FGS_FBW + 2; j += 64) b.w[j] = 0;
    wave_lds_fence();
    if (lane == 0) b.w[0] = carry;
    b.wbase = wend;
    wave_lds_fence();
}

// make room for `bits` more bits after position `bitpos`
FGI void fb_reserve(FrameBits &b, int lane, uint32_t bitpos, uint32_t bits)
{
    if (bitpos + bits - (b.wbase << 5) > 32u * FGS_FBW - 64u) fb_flush(b, lane, bitpos);
}

// ------------------------------------------------------------------ the kernel
template <bool MS, int NCH, int MAXO, bool ACC64>
__global__ void __launch_bounds__(64)
fg_encode_fast_kernel(const void *pcm, const FgBlockDesc *descs, const float *windows, FgEncParams P, uint8_t *out,
                      FgBlockResult *results, FgDebugRec *dbg, const uint16_t *crctab)
{
    constexpr int NC = MS ? 4 : NCH;
    typedef typename FastTypes<ACC64>::sum_t sum_t;
    typedef typename FastTypes<ACC64>::samp_t samp_t;
    constexpr uint32_t PADE = FastTypes<ACC64>::PADE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FgBlockDesc d = descs[blockIdx.x];
    const int lane = threadIdx.x;
    const uint32_t n = d.n;
    const uint32_t seg = n >> 6;                    // samples per lane
    const uint32_t rstr = seg + PADE;               // LDS row stride (elements)
    // ---- LDS carve
    LDS unsigned char *lbase = (LDS unsigned char *)smem;
    uint32_t off = 0;
#define FG_CARVE(type, bytes) (LDS type *)(lbase + off); off += (uint32_t)(((bytes) + 15) & ~15u)
    LDS samp_t *sL = FG_CARVE(samp_t, (P.sig_stride + 128) * sizeof(samp_t));      // 64 rows, 2 elements of skew each
    LDS samp_t *sR = FG_CARVE(samp_t, NCH == 2 ? (P.sig_stride + 128) * sizeof(samp_t) : 16);
    // the analysis scratch (autocorrelation staging rows) and the frame-bit window of the packing stage are never live
    // together: one region.  LDS is the occupancy limit (it is handed out in 1280-byte granules; 18 granules = 7 blocks
    // per CU), so the CRC tables are read from global memory (3 KB, L1-resident) instead of being copied in.
    uint32_t ubytes = NC * FGS_CSTR * 8;
    if (ubytes < (FGS_FBW + 2) * 4) ubytes = (FGS_FBW + 2) * 4;
    LDS double *dbuf = FG_CARVE(double, ubytes);
    LDS uint32_t *fbw = (LDS uint32_t *)dbuf;
    const uint16_t *crct = crctab;          // [0,256) byte table, [256,768) x^2048 tables, [1024,1792) slicing tables
    LDS double *autoc = FG_CARVE(double, NC * P.nvec * (MAXO + 1) * 8);
    LDS int32_t *qres = FG_CARVE(int32_t, NC * P.nvec * MAXO * 4);
    LDS uint32_t *lres = FG_CARVE(uint32_t, NC * P.nvec * 4);
    LDS int32_t *bestq = FG_CARVE(int32_t, NC * MAXO * 4);
    LDS uint32_t *misc = FG_CARVE(uint32_t, 128 * 4);
#undef FG_CARVE
    const float *window = windows + d.win_off;
    misc[64 + lane] = crctab[768 + lane];
    FgDebugRec *mydbg = dbg ? dbg + d.out_slot : nullptr;
#define FG_STAMP(i) do { if (mydbg && lane == 0) mydbg->t[i] = clock64(); } while (0)
    FG_STAMP(0);
    uint32_t err = 0;
    // rows of this lane and of its left neighbour (history)
    const LDS samp_t *rowL = sL + (uint32_t)lane * rstr, *rowR = sR + (uint32_t)lane * rstr;
    // g -> LDS element index: g + (g / seg) * PADE, the division by a reciprocal (exact for g < 2^16, seg <= 1024)
    const uint32_t magic = 0xFFFFFFFFu / seg + 1;
#define FG_SADDR(g) ((g) + __umulhi((g), magic) * PADE)

    // ================================================================ stage: HBM -> LDS (coalesced reads, lane-segment rows)
    {
        const int32_t lim = (int32_t)(P.bps - 1);
        uint32_t bad = 0;
        uint32_t istart = 0;
        if (NCH == 2 && !P.pcm_i16 && (d.pcm_off & 1) == 0 && (n & 127) == 0 && (((uintptr_t)pcm) & 15) == 0) {
            // int32 stereo, 16-byte aligned: two inter-channel samples per lane and load, eight loads in flight; the pair
            // lands in one LDS row (even index, even row length) and is stored as one word per channel when staged as int16
            const int4 *src = (const int4 *)((const int2 *)pcm + d.pcm_off);
            for (uint32_t j0 = 0; j0 < n / 2; j0 += 512) {
                int4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t j = j0 + u * 64 + lane;
                    v[u] = make_int4(0, 0, 0, 0);
                    if (j < n / 2) v[u] = src[j];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t j = j0 + u * 64 + lane;
                    if (j < n / 2) {
                        const int32_t l0 = v[u].x, r0 = v[u].y, l1 = v[u].z, r1 = v[u].w;
                        if (P.bps < 32) bad |= (uint32_t)(((l0 ^ (l0 >> 31)) >> lim) | ((r0 ^ (r0 >> 31)) >> lim) | ((l1 ^ (l1 >> 31)) >> lim) | ((r1 ^ (r1 >> 31)) >> lim));
                        const uint32_t ad = FG_SADDR(2 * j);
                        if (sizeof(samp_t) == 2) {
                            *(LDS uint32_t *)(sL + ad) = ((uint32_t)l0 & 0xFFFFu) | ((uint32_t)l1 << 16);
                            *(LDS uint32_t *)(sR + ad) = ((uint32_t)r0 & 0xFFFFu) | ((uint32_t)r1 << 16);
                        }
                        else { sL[ad] = (samp_t)l0; sL[ad + 1] = (samp_t)l1; sR[ad] = (samp_t)r0; sR[ad + 1] = (samp_t)r1; }
                    }
                }
            }
            istart = n;
        }
        for (uint32_t i0 = istart; i0 < n; i0 += 256) {
            int32_t a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * 64 + lane;
                a[u] = 0; b[u] = 0;
                if (i < n) {
                    if (NCH == 2) {
                        if (P.pcm_i16) { const short2 v = ((const short2 *)pcm)[d.pcm_off + i]; a[u] = v.x; b[u] = v.y; }
                        else { const int2 v = ((const int2 *)pcm)[d.pcm_off + i]; a[u] = v.x; b[u] = v.y; }
                    }
                    else {
                        if (P.pcm_i16) a[u] = ((const int16_t *)pcm)[d.pcm_off + i];
                        else a[u] = ((const int32_t *)pcm)[d.pcm_off + i];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * 64 + lane;
                if (i < n) {
                    if (P.bps < 32) bad |= (uint32_t)(((a[u] ^ (a[u] >> 31)) >> lim) | ((b[u] ^ (b[u] >> 31)) >> lim));
                    const uint32_t ad = FG_SADDR(i);
                    sL[ad] = (samp_t)a[u];
                    if (NCH == 2) sR[ad] = (samp_t)b[u];
                }
            }
        }
        if (__any(bad != 0)) err |= FG_ERR_RANGE;
        wave_lds_fence();
    }
    FG_STAMP(1);

    uint32_t pmax0 = 0;
    { uint32_t b = n; while (!(b & 1)) { pmax0++; b >>= 1; } if (pmax0 > 15) pmax0 = 15; }
    if (P.max_po < pmax0) pmax0 = P.max_po;
    const uint32_t pmin0 = P.min_po < pmax0 ? P.min_po : pmax0;

    // ================================================================ wasted bits + fixed-predictor error sums (one pass)
    // The per-lane sums double as the Rice partition sums of the fixed predictors (the order-k residual IS the k-th
    // difference), so the fixed predictor needs no FIR pass later: facc = sums over the lane's samples from sample 4 on
    // (what libFLAC's order guess uses), fwarm = the part of samples 0..3 that belongs to the order-k residual (s >= k).
    uint32_t wst[NC], sbp[NC];
    u64 tot[NC][5];
    sum_t facc[NC][5], fwarm[NC][5];
    {
        uint32_t orv[NC];
        int32_t p1[NC], q1[NC], q2[NC], q3[NC];       // previous value and previous 1st..3rd differences
#pragma unroll
        for (int c = 0; c < NC; c++) {
            orv[c] = 0; p1[c] = 0; q1[c] = 0; q2[c] = 0; q3[c] = 0;
#pragma unroll
            for (int kk = 0; kk < 5; kk++) { facc[c][kk] = 0; fwarm[c][kk] = 0; }
        }
        // head: prime the differences with the four samples in front of the segment (zeros in front of the block), then
        // samples 0..3, which only lanes > 0 add to the sums (libFLAC hands fixed.c the signal shifted by the maximum
        // fixed order, so the sums of the block start at sample 4)
#pragma unroll
        for (int s = -4; s < 4; s++) {
            int32_t l = 0, r = 0;
            if (s >= 0) { l = rowL[s]; r = (NCH == 2) ? rowR[s] : 0; }
            else if (lane > 0) { l = rowL[(int)seg + s - (int)rstr]; r = (NCH == 2) ? rowR[(int)seg + s - (int)rstr] : 0; }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                int32_t v;
                if (!MS) v = (c == 0) ? l : r;
                else v = (c == 0) ? l : (c == 1) ? r : (c == 2) ? ((l + r) >> 1) : (l - r);
                const int32_t e1 = v - p1[c], e2 = e1 - q1[c], e3 = e2 - q2[c], e4 = e3 - q3[c];
                p1[c] = v; q1[c] = e1; q2[c] = e2; q3[c] = e3;
                if (s >= 0) {
                    orv[c] |= (uint32_t)v;
                    const uint32_t ab[5] = {fabs32(v), fabs32(e1), fabs32(e2), fabs32(e3), fabs32(e4)};
#pragma unroll
                    for (int kk = 0; kk < 5; kk++) {
                        facc[c][kk] += (lane > 0) ? ab[kk] : 0u;
                        if (s >= kk) fwarm[c][kk] += ab[kk];
                    }
                }
            }
        }
#pragma unroll 1
        for (int s = 4; s < (int)seg; s++) {
            const int32_t l = rowL[s], r = (NCH == 2) ? rowR[s] : 0;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                int32_t v;
                if (!MS) v = (c == 0) ? l : r;
                else v = (c == 0) ? l : (c == 1) ? r : (c == 2) ? ((l + r) >> 1) : (l - r);
                const int32_t e1 = v - p1[c], e2 = e1 - q1[c], e3 = e2 - q2[c], e4 = e3 - q3[c];
                p1[c] = v; q1[c] = e1; q2[c] = e2; q3[c] = e3;
                orv[c] |= (uint32_t)v;
                facc[c][0] += fabs32(v); facc[c][1] += fabs32(e1); facc[c][2] += fabs32(e2);
                facc[c][3] += fabs32(e3); facc[c][4] += fabs32(e4);
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint32_t o = wave_or32(orv[c]);
            uint32_t w = o ? (uint32_t)__builtin_ctz(o) : 0;
            const uint32_t nominal = P.bps + ((MS && c == 3) ? 1u : 0u);
            if (w > nominal) w = nominal;
            wst[c] = w; sbp[c] = nominal - w;
#pragma unroll
            for (int kk = 0; kk < 5; kk++) tot[c][kk] = ACC64 ? wave_sum64((u64)facc[c][kk]) : (u64)wave_sum((uint32_t)facc[c][kk]);
        }
        // Blocks in which some candidate has wasted bits (all samples share trailing zero bits: rare) are handed
        // to the generic kernel: every hot loop below then works on unshifted samples.
        {
            uint32_t anyw = 0;
#pragma unroll
            for (int c = 0; c < NC; c++) anyw |= wst[c];
            if (anyw) {
                if (lane == 0) {
                    FgBlockResult *r = &results[d.out_slot];
                    r->bytes = 0; r->ca = 0; r->err = FG_ERR_REDO; r->reserved = 2;
                }
                return;
            }
        }
    }

    // ---- per-candidate baseline: verbatim / constant, fixed order guess
    uint32_t best[NC], guess[NC];
    uint32_t d_type[NC], d_order[NC], d_prec[NC], d_porder[NC], d_method[NC], d_k[NC];
    int d_shift[NC];
    uint32_t fixed_mask = 0, lpc_mask = 0;
    bool lmb_forced = false;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t sb = sbp[c];
        const u64 vb = (u64)8 + (u64)n * sb;
        best[c] = vb < 0xFFFFFFFFull ? (uint32_t)vb : 0xFFFFFFFFu;
        d_type[c] = 1; d_order[c] = 0; d_prec[c] = 0; d_shift[c] = 0; d_porder[c] = 0; d_method[c] = 0; d_k[c] = 0;
        const u64 m34 = tot[c][3] < tot[c][4] ? tot[c][3] : tot[c][4];
        const u64 m234 = tot[c][2] < m34 ? tot[c][2] : m34;
        const u64 m1234 = tot[c][1] < m234 ? tot[c][1] : m234;
        uint32_t g;
        u64 tg;
        if (tot[c][0] <= m1234) { g = 0; tg = tot[c][0]; }
        else if (tot[c][1] <= m234) { g = 1; tg = tot[c][1]; }
        else if (tot[c][2] <= m34) { g = 2; tg = tot[c][2]; }
        else if (tot[c][3] <= tot[c][4]) { g = 3; tg = tot[c][3]; }
        else { g = 4; tg = tot[c][4]; }
        guess[c] = g;
        const double len = (double)(n - 4);
        const float rbg = (float)((tg > 0) ? log(FG_LN2 * (double)tg / len) / FG_LN2 : 0.0);
        bool constant = false;
        if (tot[c][1] == 0) {
            const int32_t l0 = sL[0], r0 = (NCH == 2) ? sR[0] : 0;
            const int32_t x0 = !MS ? (c == 0 ? l0 : r0) : (c == 0 ? l0 : c == 1 ? r0 : c == 2 ? ((l0 + r0) >> 1) : (l0 - r0));
            uint32_t ne = 0;
#pragma unroll 1
            for (uint32_t s = 0; s < seg; s++) {
                const int32_t l = rowL[s], r = (NCH == 2) ? rowR[s] : 0;
                const int32_t x = !MS ? (c == 0 ? l : r) : (c == 0 ? l : c == 1 ? r : c == 2 ? ((l + r) >> 1) : (l - r));
                ne |= (x != x0);
            }
            constant = !__any(ne != 0);
        }
        if (mydbg && lane == 0) {
            for (int kk = 0; kk < 5; kk++) mydbg->cand[c].fixed_tot[kk] = tot[c][kk];
            mydbg->cand[c].fixed_guess = g;
        }
        // limit_min_bitrate (libFLAC 1.4.3 as observed, oracle/flac_oracle.c): the last independent channel is evaluated with
        // CONSTANT disabled when every earlier one chose CONSTANT; mid and side of the same frame then are, too -- unless
        // only mid/side are evaluated at all (loose mid-side follower frames, forced_ca == 3)
        if (P.limit_min_bitrate) {
            bool forbid = false;
            if (c < NCH) {
                if (c == NCH - 1 && d.forced_ca != 3) {
                    forbid = true;
#pragma unroll
                    for (int cc = 0; cc < NCH - 1; cc++) if (d_type[cc] != 0) forbid = false;
                    lmb_forced = forbid;
                }
            }
            else forbid = lmb_forced;
            if (forbid) constant = false;
        }
        if (constant) {
            const uint32_t cb = 8 + sb;
            if (cb < best[c]) { best[c] = cb; d_type[c] = 0; }
        }
        else {
            if (!(rbg >= (float)sb)) fixed_mask |= 1u << c;
            if (P.max_lpc_order > 0) lpc_mask |= 1u << c;
        }
    }
    FG_STAMP(2);

    // ================================================================ autocorrelation vectors (order-preserving fp64 chains)
    // lane = candidate row (16 lanes) x lag.  Per chunk of FGS_CK samples the windowed signal of every candidate is staged
    // as doubles (with FGS_DH history entries in front); the chain then needs one LDS read per step: lane (c, l) reads
    // d[j - l], the d[j] operand is lane (c, 0)'s own value, broadcast inside the FMA (DPP row_newbcast:0).
    uint32_t nv = 0;
    const uint32_t mo = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
    if (lpc_mask && MAXO > 0 && mo > 0) {
        const uint32_t cl = lane >> 4, l = lane & 15;
        const bool on = cl < (uint32_t)NC && l <= mo;
        const LDS double *hist = dbuf + (cl < (uint32_t)NC ? cl : 0) * FGS_CSTR + FGS_DH - (on ? l : 0);
        // vector schedule of apply_apodization_ (tukey: one; subdivide_tukey(parts): whole, then per depth b the
        // partial windows at even c and the punch-outs at odd c)
        uint32_t vb_ = 1, vc_ = 0;
        bool more = true;
        while (more) {
            uint32_t vec_len = n, part = 0, sh = 0;
            bool punch = false, skip = false;
            if (nv > 0) {
                if (n / vb_ <= 32) skip = true;
                else if (!(vc_ & 1)) { vec_len = n / vb_; part = n / vb_ / 2; sh = (vc_ / 2 * n) / vb_; }
                else punch = true;
            }
            if (!skip && !punch) {
                double acc = 0.0;
                for (uint32_t j = lane; j < NC * FGS_DH; j += 64) dbuf[(j / FGS_DH) * FGS_CSTR + (j % FGS_DH)] = 0.0;
                wave_lds_fence();
                // window values of the first chunk
                float wv[(FGS_CK + 63) / 64];
                uint32_t si[(FGS_CK + 63) / 64];
                auto fetch = [&](uint32_t k0) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < (FGS_CK + 63) / 64; u++) {
                        const uint32_t i = k0 + u * 64 + lane;
                        float w = 0.0f;
                        uint32_t s_ = 0;
                        if (i < vec_len && (uint32_t)(u * 64 + lane) < FGS_CK) {
                            if (part == 0) { w = window[i]; s_ = i; }
                            else if (i < part) { w = window[i]; s_ = sh + i; }
                            else if (i < 2 * part) { w = window[n - 2 * part + i]; s_ = sh + i; }
                        }
                        wv[u] = w; si[u] = s_;
                    }
                };
                fetch(0);
                for (uint32_t k0 = 0; k0 < vec_len; k0 += FGS_CK) {
                    const uint32_t kn = (vec_len - k0) < FGS_CK ? (vec_len - k0) : FGS_CK;
#pragma unroll
                    for (int u = 0; u < (FGS_CK + 63) / 64; u++) {
                        const uint32_t j = u * 64 + lane;
                        if (j < kn) {
                            const uint32_t ad = FG_SADDR(si[u]);
                            const int32_t L = sL[ad], R = (NCH == 2) ? sR[ad] : 0;
                            const bool zero = part != 0 && (k0 + j) >= 2 * part;
#pragma unroll
                            for (int c = 0; c < NC; c++) {
                                const int32_t x = !MS ? (c == 0 ? L : R) : (c == 0 ? L : c == 1 ? R : c == 2 ? ((L + R) >> 1) : (L - R));
                                const float dd = zero ? 0.0f : (float)x * wv[u];
                                dbuf[c * FGS_CSTR + FGS_DH + j] = (double)dd;
                            }
                        }
                    }
                    if (k0 + FGS_CK < vec_len) fetch(k0 + FGS_CK);       // next chunk's window values travel during the chain
                    wave_lds_fence();
                    if (on) {
                        // four steps per group; the operands of the next two groups are requested before the FMAs of the
                        // current one so that the LDS latency overlaps the dependent chain (reads past the chunk stay
                        // inside the row and are not used)
#define FG_FMAC4(h) asm("v_fmac_f64_dpp %0, %1, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"            \
                        "v_fmac_f64_dpp %0, %2, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"            \
                        "v_fmac_f64_dpp %0, %3, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"            \
                        "v_fmac_f64_dpp %0, %4, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf"                 \
                        : "+v"(acc) : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]))
#define FG_LOAD4(h, j0) do { h[0] = hist[(j0)]; h[1] = hist[(j0) + 1]; h[2] = hist[(j0) + 2]; h[3] = hist[(j0) + 3]; } while (0)
                        double ha[4], hb[4], hc[4];
                        uint32_t j = 0;
                        FG_LOAD4(ha, 0); FG_LOAD4(hb, 4);
                        for (; j + 12 <= kn; j += 12) {
                            FG_LOAD4(hc, j + 8);  FG_FMAC4(ha);
                            FG_LOAD4(ha, j + 12); FG_FMAC4(hb);
                            FG_LOAD4(hb, j + 16); FG_FMAC4(hc);
                        }
                        if (j + 4 <= kn) { FG_FMAC4(ha); j += 4; if (j + 4 <= kn) { FG_FMAC4(hb); j += 4; } }
                        for (; j < kn; j++) {
                            const double h0 = hist[j];
                            asm("v_fmac_f64_dpp %0, %1, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h0));
                        }
#undef FG_FMAC4
#undef FG_LOAD4
                    }
                    wave_lds_fence();
                    if (k0 + kn < vec_len) {
                        // the last FGS_DH entries of the chunk become the history of the next one
                        double t[(NC * FGS_DH + 63) / 64];
#pragma unroll
                        for (int u = 0; u < (NC * FGS_DH + 63) / 64; u++) {
                            const uint32_t j = u * 64 + lane;
                            t[u] = (j < NC * FGS_DH) ? dbuf[(j / FGS_DH) * FGS_CSTR + FGS_CK + (j % FGS_DH)] : 0.0;
                        }
                        wave_lds_fence();
#pragma unroll
                        for (int u = 0; u < (NC * FGS_DH + 63) / 64; u++) {
                            const uint32_t j = u * 64 + lane;
                            if (j < NC * FGS_DH) dbuf[(j / FGS_DH) * FGS_CSTR + (j % FGS_DH)] = t[u];
                        }
                        wave_lds_fence();
                    }
                }
                if (on) autoc[(cl * P.nvec + nv) * (MAXO + 1) + l] = acc;
                wave_lds_fence();
            }
            else if (punch) {
                // root - previous partial for lags < mo; lag mo keeps the partial (upstream quirk)
                const uint32_t total = (uint32_t)NC * (mo + 1);
                for (uint32_t j = lane; j < total; j += 64) {
                    const uint32_t c = j / (mo + 1), ll = j % (mo + 1);
                    LDS double *base = autoc + c * P.nvec * (MAXO + 1);
                    const double prev = base[(nv - 1) * (MAXO + 1) + ll];
                    base[nv * (MAXO + 1) + ll] = (ll < mo) ? base[ll] - prev : prev;
                }
                wave_lds_fence();
            }
            if (!skip) nv++;
            // next vector
            if (P.apod_parts < 2) more = false;
            else if (nv == 1 && vb_ == 1) { vb_ = 2; vc_ = 0; }
            else {
                if (vb_ == 2) { if (vc_ == 0) vc_ = 2; else { vc_ = 0; vb_++; } }
                else if (vc_ < 2 * vb_ - 1) vc_++;
                else { vc_ = 0; vb_++; }
                if (vb_ > P.apod_parts) more = false;
            }
        }
        if (mydbg) {
            for (uint32_t j = lane; j < (uint32_t)NC * nv * (mo + 1); j += 64) {
                const uint32_t c = j / (nv * (mo + 1)), r = j % (nv * (mo + 1)), v = r / (mo + 1), ll = r % (mo + 1);
                mydbg->cand[c].autoc[v][ll] = autoc[(c * P.nvec + v) * (MAXO + 1) + ll];
            }
            if (lane < NC) mydbg->cand[lane].nvec = nv;
        }
    }
    FG_STAMP(4);

    // ================================================================ Levinson-Durbin, order guess, quantiser
    // lane = candidate * nvec + vector.  lres[idx] = order | prec<<8 | (shift&255)<<16 | ok<<24 | ran<<25
    // The recursion runs in registers (loops unrolled to the compile-time maximum order, guarded by the run-time one);
    // the coefficient set of the best order so far is kept aside instead of being recomputed afterwards (same operations
    // in the same order as lpc.c, so the same doubles).
    if (nv > 0) {
        const uint32_t nidx = (uint32_t)NC * P.nvec;
        const uint32_t idx = lane;
        if (idx < nidx) {
            const uint32_t c = idx / P.nvec, v = idx % P.nvec;
            const LDS double *Ap = autoc + (c * P.nvec + v) * (MAXO + 1);
            bool on = v < nv && ((lpc_mask >> c) & 1);
            double A[MAXO + 1];
#pragma unroll
            for (int j = 0; j <= MAXO; j++) A[j] = ((uint32_t)j <= mo) ? Ap[j] : 0.0;
            if (on && A[0] == 0.0) on = false;
            if (!on) {
#pragma unroll
                for (int j = 0; j <= MAXO; j++) A[j] = 0.0;
            }
            uint32_t sb = sbp[0];
#pragma unroll
            for (int cc = 1; cc < NC; cc++) if (c == (uint32_t)cc) sb = sbp[cc];
            const double a0 = on ? A[0] : 1.0;
            const uint32_t overhead = sb + P.qlp_precision;
            const double scale = 0.5 / (double)n;
            double er = a0, bestb = 4294967295.0;
            double lp[MAXO], keep[MAXO], err2 = a0;
#pragma unroll
            for (int j = 0; j < MAXO; j++) { lp[j] = 0.0; keep[j] = 0.0; }
            uint32_t besti = 0;
            bool stopped = false, have = false;
#pragma unroll
            for (int i = 0; i < MAXO; i++) {
                if ((uint32_t)i < mo) {
                    double r = -A[i + 1];
#pragma unroll
                    for (int j = 0; j < i; j++) r -= lp[j] * A[i - j];
                    r /= er;
                    lp[i] = r;
#pragma unroll
                    for (int j = 0; j < (i >> 1); j++) {
                        const double tmp = lp[j], t2 = lp[i - 1 - j];
                        lp[j] = tmp + r * t2;
                        lp[i - 1 - j] = t2 + r * tmp;
                    }
                    if (i & 1) { const double t = lp[i >> 1]; lp[i >> 1] = t + t * r; }
                    er *= (1.0 - r * r);
                    bool better = false;
                    if (!stopped) {
                        const uint32_t o = i + 1;
                        const double bits = f_ebps(er, scale) * (double)(n - o) + (double)(o * overhead);
                        if (bits < bestb) { besti = i; bestb = bits; better = true; }
                        if (er == 0.0) stopped = true;
                    }
                    // order 1 is what remains when no order improves on the initial bound (besti stays 0)
                    if (better || !have) {
                        if (better || i == 0) {
#pragma unroll
                            for (int j = 0; j <= i; j++) keep[j] = lp[j];
                            err2 = er;
                        }
                        have = true;
                    }
                }
            }
            const uint32_t ostar = besti + 1;
            float lpf[MAXO];
#pragma unroll
            for (int j = 0; j < MAXO; j++) lpf[j] = (float)(-keep[j]);
            uint32_t result = 0;
            for (uint32_t j = 0; j < (uint32_t)MAXO; j++) qres[idx * MAXO + j] = 0;
            if (on) {
                bool ok = !(f_ebps(err2, 0.5 / (double)(n - ostar)) >= (double)sb);
                uint32_t prec = P.qlp_precision;
                if (sb <= 17) { const uint32_t lim = 32 - sb - ilog2_32(ostar); if (lim < prec) prec = lim; }
                int shift = 0;
                if (ok) {
                    const int p1 = (int)prec - 1;
                    const int32_t qmax = (1 << p1) - 1, qmin = -(1 << p1);
                    double cmax = 0.0;
#pragma unroll
                    for (int j = 0; j < MAXO; j++) if ((uint32_t)j < ostar) { const double dd = fabs((double)lpf[j]); if (dd > cmax) cmax = dd; }
                    if (cmax <= 0.0) ok = false;
                    else {
                        const int e = (int)((__double_as_longlong(cmax) >> 52) & 0x7FF) - 1022;
                        shift = p1 - (e - 1) - 1;
                        if (shift > 15) shift = 15;
                        else if (shift < -16) ok = false;
                    }
                    if (ok) {
                        double error = 0.0;
                        const bool neg = shift < 0;
                        const double mul = neg ? (double)(1 << (-shift)) : (double)(1 << shift);
#pragma unroll
                        for (int j = 0; j < MAXO; j++) {
                            if ((uint32_t)j < ostar) {
                                const double lpv = (double)lpf[j];
                                error += neg ? lpv / mul : lpv * mul;
                                const double rq = round(error);
                                int32_t qv = (int32_t)(i64)rq;
                                if (qv > qmax) qv = qmax; else if (qv < qmin) qv = qmin;
                                error -= (double)qv;
                                qres[idx * MAXO + j] = qv;
                            }
                        }
                        if (neg) shift = 0;
                    }
                }
                result = ostar | (prec << 8) | (((uint32_t)shift & 0xFF) << 16) | ((ok ? 1u : 0u) << 24) | (1u << 25);
            }
            lres[idx] = result;
        }
        wave_lds_fence();
    }
    FG_STAMP(5);

    // ================================================================ evaluation of the predictors: pass 0 = fixed, then one
    // pass per autocorrelation vector.  Each pass: every lane runs the FIR of every candidate over its segment (history in
    // registers, coefficients in SGPRs); its |residual| total is a partition sum.  Then the Rice parameter / partition
    // order search (lane = partition) and the strict-< update of the best (libFLAC's candidate order).
    u64 te_fir = 0, te_search = 0, te_setup = 0, te_l = mydbg ? clock64() : 0;
#define FG_TE(acc) do { if (mydbg) { const u64 n_ = clock64(); acc += n_ - te_l; te_l = n_; } } while (0)
#pragma unroll 1
    for (uint32_t pass = 0; pass < 1 + nv; pass++) {
        uint32_t order[NC], prec[NC], emask = 0;
        int32_t q[NC][MAXO];
        int shift[NC];
        const int kind = pass == 0 ? 0 : 1;
        if (pass == 0) {
            if (!fixed_mask) continue;
            emask = fixed_mask;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const uint32_t g = guess[c];
                order[c] = g; shift[c] = 0; prec[c] = 0;
                // binomial coefficients of the order-g difference
                const int32_t c0 = g == 0 ? 0 : (int32_t)g, c1 = g < 2 ? 0 : (g == 2 ? -1 : g == 3 ? -3 : -6);
                const int32_t c2 = g < 3 ? 0 : (g == 3 ? 1 : 4), c3 = g < 4 ? 0 : -1;
#pragma unroll
                for (int j = 0; j < MAXO; j++) q[c][j] = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : j == 3 ? c3 : 0;
            }
        }
        else {
            const uint32_t v = pass - 1;
            // one LDS read brings every coefficient of the pass (lane = candidate * MAXO + tap), another the result words;
            // the wave-uniform copies come from v_readlane
            const uint32_t lc = (uint32_t)lane / MAXO, lj = (uint32_t)lane % MAXO;
            const int32_t qall = (lc < (uint32_t)NC) ? qres[(lc * P.nvec + v) * MAXO + lj] : 0;
            const uint32_t rall = ((uint32_t)lane < (uint32_t)NC) ? lres[(uint32_t)lane * P.nvec + v] : 0;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const uint32_t r = rl(rall, c);
                order[c] = r & 0xFF; prec[c] = (r >> 8) & 0xFF; shift[c] = (int)(int8_t)((r >> 16) & 0xFF);
                if (((lpc_mask >> c) & 1) && ((r >> 24) & 1)) emask |= 1u << c;
                if (order[c] == 0) order[c] = 1;
#pragma unroll
                for (int j = 0; j < MAXO; j++) q[c][j] = (int32_t)rl((uint32_t)qall, c * MAXO + j);
                if (mydbg && lane == 0) mydbg->cand[c].lpc_guess[v] = ((r >> 25) & 1) ? (r & 0xFF) : 0;
            }
            if (!emask) continue;
        }
        FG_TE(te_setup);
        FG_TE(te_setup);
        sum_t psum[NC];
        uint32_t ovf[NC];
        if (pass == 0) {
            // fixed predictor of order g: the lane's partition sum is its sum of |g-th differences|, lane 0 adds the part
            // of samples g..3
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const uint32_t g = guess[c];
                sum_t a = g == 0 ? facc[c][0] : g == 1 ? facc[c][1] : g == 2 ? facc[c][2] : g == 3 ? facc[c][3] : facc[c][4];
                const sum_t w = g == 0 ? fwarm[c][0] : g == 1 ? fwarm[c][1] : g == 2 ? fwarm[c][2] : g == 3 ? fwarm[c][3] : fwarm[c][4];
                if (lane == 0) a += w;
                psum[c] = a; ovf[c] = 0;
            }
        }
        else {
            // ---- FIR over the segment.  The coefficients live in VGPRs (pinned by an opaque asm): 32 wave-uniform values
            // plus the rest of the uniform state overflow the SGPR file.
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int j = 0; j < MAXO; j++) asm volatile("" : "+v"(q[c][j]));
            int32_t h[NC][MAXO];
#pragma unroll
            for (int c = 0; c < NC; c++) { psum[c] = 0; ovf[c] = 0; }
            // history: the MAXO samples in front of the segment; sample (s - 1 - j) lives in slot (s - 1 - j) mod MAXO
#pragma unroll
            for (int j = 0; j < MAXO; j++) {
                int32_t l = 0, r = 0;
                if (lane > 0) { l = rowL[(int)seg - 1 - j - (int)rstr]; r = (NCH == 2) ? rowR[(int)seg - 1 - j - (int)rstr] : 0; }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const int32_t x = !MS ? (c == 0 ? l : r) : (c == 0 ? l : c == 1 ? r : c == 2 ? ((l + r) >> 1) : (l - r));
                    h[c][(MAXO - 1 - j) % MAXO] = ACC64 ? fpack(x) : x;
                }
            }
            auto step = [&](int u, uint32_t s, bool guard) __attribute__((always_inline)) {
                const int32_t l = rowL[s], r = (NCH == 2) ? rowR[s] : 0;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const int32_t x = !MS ? (c == 0 ? l : r) : (c == 0 ? l : c == 1 ? r : c == 2 ? ((l + r) >> 1) : (l - r));
                    int32_t res;
                    if (!ACC64) res = x - (fir24<MAXO>(q[c], h[c], u) >> shift[c]);
                    else {
                        const i64 rr = (i64)x - (fir48<MAXO>(q[c], h[c], u) >> shift[c]);
                        if (rr <= (i64)INT32_MIN || rr > (i64)INT32_MAX) ovf[c] = 1;
                        res = (int32_t)rr;
                    }
                    h[c][u] = ACC64 ? fpack(x) : x;
                    // warm-up samples (the first `order` of the block, all in lane 0) are not residuals
                    if (!guard || lane > 0 || s >= order[c]) psum[c] += fabs32(res);
                }
            };
            uint32_t s0 = 0;
            // first group: may contain warm-up samples
            if (seg >= (uint32_t)MAXO) {
#pragma unroll
                for (int u = 0; u < MAXO; u++) step(u, (uint32_t)u, true);
                s0 = MAXO;
            }
#pragma unroll 1
            for (; s0 + MAXO <= seg; s0 += MAXO) {
#pragma unroll
                for (int u = 0; u < MAXO; u++) step(u, s0 + u, false);
            }
#pragma unroll
            for (int u = 0; u < MAXO; u++) if (s0 + u < seg) step(u, s0 + u, s0 == 0);
        }
        FG_TE(te_fir);
        // ---- Rice parameter / partition order search, all candidates side by side.  The sums stay where they are:
        // at partition order po, partition p lives in lane p * (64 >> po); going one order down adds the neighbour
        // 2^(5-po) lanes up (DPP row shift inside a 16-lane row, a permute across rows).
        {
            u64 sv[NC];
            uint32_t best_bits[NC], bpo[NC], kb[NC];
            bool dead[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                sv[c] = (u64)psum[c]; best_bits[c] = 0; bpo[c] = 0; kb[c] = 0;
                dead[c] = ACC64 && __any(ovf[c] != 0);
            }
            // lane + dist for dist = 1, 2, 4, 8 (zero past the end of the row), 16, 32
            auto up = [&](uint32_t v, uint32_t t) __attribute__((always_inline)) -> uint32_t {
                switch (t) {
                case 0: return dpp0<0x101>(v);
                case 1: return dpp0<0x102>(v);
                case 2: return dpp0<0x104>(v);
                case 3: return dpp0<0x108>(v);
                case 4: return (uint32_t)__shfl((int)v, (lane + 16) & 63);
                default: return (uint32_t)__shfl((int)v, (lane + 32) & 63);
                }
            };
            auto merge = [&](uint32_t t) __attribute__((always_inline)) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    u64 o = up((uint32_t)sv[c], t);
                    if (ACC64) o |= (u64)up((uint32_t)(sv[c] >> 32), t) << 32;
                    sv[c] += o;
                }
            };
            for (uint32_t m = 6; m > pmax0; m--) merge(6 - m);
            const uint32_t psz0 = n >> pmax0;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const bool wrap32 = (sbp[c] + 4) < (32 - ilog2_32(psz0));
                if (wrap32) sv[c] &= 0xFFFFFFFFull;
            }
            const uint32_t limit = P.rice_limit;
            // 0x40000 / x for x < 2^16 through the reciprocal, corrected to the exact quotient
            auto div18 = [&](uint32_t x) __attribute__((always_inline)) -> uint32_t {
                uint32_t qd = (uint32_t)(262144.0f * __builtin_amdgcn_rcpf((float)x));
                const int32_t r = (int32_t)(0x40000u - qd * x);
                if (r < 0) qd--;
                else if ((uint32_t)r >= x) qd++;
                return qd;
            };
            const bool lane0 = lane == 0;
            // one partition order; PO is a compile-time constant so that the lane stride and the DPP shifts are immediates
            auto po_step = [&](auto PO) __attribute__((always_inline)) {
                constexpr int po = decltype(PO)::value;
                if (po > (int)pmax0 || po < (int)pmin0) return;
                constexpr uint32_t stride = 64u >> po;
                const bool valid = ((uint32_t)lane & (stride - 1)) == 0;
                const uint32_t pbase = n >> po;
                const uint32_t dv_all = div18(pbase);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const uint32_t np = lane0 ? pbase - order[c] : pbase;
                    const uint32_t dv0 = div18(pbase - order[c]);
                    const uint32_t dv = lane0 ? dv0 : dv_all;
                    uint32_t kr, bits;
                    if (!ACC64) {
                        // sums below 2^31 (16-bit input): 32-bit arithmetic, one 64-bit product
                        const uint32_t s32 = (uint32_t)sv[c];
                        const uint32_t s1 = (s32 > 1 ? s32 : 1) - 1;
                        const uint32_t qv = (uint32_t)(((u64)s1 * dv) >> 18);
                        kr = qv ? 32 - (uint32_t)__builtin_clz(qv) : 0;
                        if (kr >= limit) kr = limit - 1;
                        uint32_t pb = 4 + (1 + kr) * np + ((s32 << 1) >> kr) - (np >> 1);
                        if (!valid) pb = 0;
                        u64 total;
                        if (__any(pb >> 25)) total = wave_sum64((u64)pb) + 6;
                        else total = (u64)wave_sum(pb) + 6;
                        bits = total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total;
                    }
                    else {
                        const u64 s = sv[c];
                        kr = 0;
                        if (s >= 2) {
                            const u64 qv = ((s - 1) * dv) >> 18;
                            if (qv != 0) kr = ilog2_64(qv) + 1;
                        }
                        if (kr >= limit) kr = limit - 1;
                        u64 pb = (u64)4 + (u64)(1 + kr) * np + (kr ? (s >> (kr - 1)) : (s << 1)) - (np >> 1);
                        if (pb > 0xFFFFFFFFull) pb = 0xFFFFFFFFull;
                        if (!valid) pb = 0;
                        u64 total;
                        if (__any(pb >> 25)) total = wave_sum64(pb) + 6;
                        else total = (u64)wave_sum((uint32_t)pb) + 6;
                        bits = total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total;
                    }
                    if (best_bits[c] == 0 || bits < best_bits[c]) { best_bits[c] = bits; bpo[c] = (uint32_t)po; kb[c] = kr; }
                }
                if (po > (int)pmin0) merge(6 - (uint32_t)po);
            };
            po_step(std::integral_constant<int, 6>()); po_step(std::integral_constant<int, 5>());
            po_step(std::integral_constant<int, 4>()); po_step(std::integral_constant<int, 3>());
            po_step(std::integral_constant<int, 2>()); po_step(std::integral_constant<int, 1>());
            po_step(std::integral_constant<int, 0>());
#pragma unroll
            for (int c = 0; c < NC; c++) {
                uint32_t est = 0;
                if (((emask >> c) & 1) && !dead[c]) {
                    const uint32_t sb = sbp[c];
                    est = kind == 0 ? (8 + order[c] * sb) : (8 + 4 + 5 + order[c] * (prec[c] + sb));
                    if (best_bits[c] < 0xFFFFFFFFu - est) est += best_bits[c]; else est = 0xFFFFFFFFu;
                    if (est > 0 && est < best[c]) {
                        best[c] = est;
                        d_type[c] = kind == 0 ? 2 : 3; d_order[c] = order[c]; d_prec[c] = prec[c]; d_shift[c] = shift[c];
                        d_porder[c] = bpo[c]; d_k[c] = kb[c];          // parameter of partition p in lane p * (64 >> order)
                        d_method[c] = __any((((uint32_t)lane & ((64u >> bpo[c]) - 1)) == 0) && kb[c] >= 15) ? 1 : 0;
                        if (kind == 1 && lane < MAXO) bestq[c * MAXO + lane] = qres[((uint32_t)c * P.nvec + (pass - 1)) * MAXO + lane];
                    }
                }
                if (mydbg && lane == 0 && ((emask >> c) & 1)) {
                    if (kind == 0) mydbg->cand[c].fixed_bits = est;
                    else mydbg->cand[c].lpc_bits[pass - 1] = est;
                }
            }
        }
        wave_lds_fence();
    }
    FG_TE(te_search);
    if (mydbg && lane == 0) { mydbg->t[13] = te_fir; mydbg->t[14] = te_search; mydbg->t[15] = te_setup; }
    FG_STAMP(6);

    // ================================================================ channel assignment
    uint32_t ca = 0, sub0 = 0, sub1 = 1;
    if (MS) {
        if (d.forced_ca != 0xFF) ca = d.forced_ca & 0x7F;       // (bit 7: a loose mid-side DECISION frame, see FgBlockDesc)
        else {
            const uint32_t b01 = best[0] + best[NC > 1 ? 1 : 0], b03 = best[0] + best[NC > 3 ? 3 : 0];
            const uint32_t b13 = best[NC > 1 ? 1 : 0] + best[NC > 3 ? 3 : 0], b23 = best[NC > 2 ? 2 : 0] + best[NC > 3 ? 3 : 0];
            uint32_t mn = b01;
            if (b03 < mn) { mn = b03; ca = 1; }
            if (b13 < mn) { mn = b13; ca = 2; }
            if (b23 < mn) { mn = b23; ca = 3; }
        }
        sub0 = ca == 2 ? 3 : (ca == 3 ? 2 : 0);
        sub1 = ca == 0 ? 1 : (ca == 2 ? 1 : 3);
    }
    if (mydbg) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (lane == 0) {
                FgDebugCand *dc = &mydbg->cand[c];
                dc->wasted = wst[c]; dc->sbps = sbp[c]; dc->type = d_type[c]; dc->order = d_type[c] >= 2 ? d_order[c] : 0;
                dc->precision = d_type[c] == 3 ? d_prec[c] : 0; dc->shift = d_type[c] == 3 ? d_shift[c] : 0;
                dc->bits = best[c]; dc->porder = d_type[c] >= 2 ? d_porder[c] : 0; dc->rice_method = d_type[c] >= 2 ? d_method[c] : 0;
                for (uint32_t j = 0; j < FG_MAX_ORDER; j++) dc->qlp[j] = (d_type[c] == 3 && j < d_order[c] && j < (uint32_t)MAXO) ? bestq[c * MAXO + j] : 0;
            }
            if (d_type[c] >= 2 && ((uint32_t)lane & ((64u >> d_porder[c]) - 1)) == 0) mydbg->cand[c].rice_params[(uint32_t)lane >> (6 - d_porder[c])] = d_k[c];
        }
    }
    FG_STAMP(7);

    // ================================================================ pack: header, subframes, padding, CRC-16
    u64 tacc[4] = {0, 0, 0, 0}, tl_ = mydbg ? clock64() : 0;
#define FG_TACC(i) do { if (mydbg) { const u64 n_ = clock64(); tacc[i] += n_ - tl_; tl_ = n_; } } while (0)
    FrameBits fb;
    // (the cast only matters where this header is included with LDS-resident tables, i.e. never for this kernel)
#define FG_TP(x) ((const FG_CRCT uint16_t *)(uintptr_t)(x))
    fb.w = fbw; fb.t0 = FG_TP(crct); fb.thi = FG_TP(crct + 256); fb.tlo = FG_TP(crct + 512);
    fb.t1 = FG_TP(crct + 1024); fb.t2 = FG_TP(crct + 1280); fb.t3 = FG_TP(crct + 1536);
#undef FG_TP
    fb.outw = (uint32_t *)(out + (size_t)d.out_slot * P.slot_bytes);
    fb.slot_words = P.slot_bytes / 4; fb.wbase = 0; fb.err = 0; fb.crc = 0;
    uint32_t bitpos = 0;
    for (uint32_t j = lane; j < FGS_FBW + 2; j += 64) fbw[j] = 0;
    wave_lds_fence();
    {   // frame header: assembled by lane 0 in LDS, emitted one byte per lane
        LDS uint8_t *hb = (LDS uint8_t *)misc;
        uint32_t hl = 0;
        if (lane == 0) {
            uint32_t u, bs_hint = 0, sr_hint = 0;
            hb[hl++] = 0xFF; hb[hl++] = 0xF8;
            switch (n) {
            case 192: u = 1; break; case 576: u = 2; break; case 1152: u = 3; break; case 2304: u = 4; break;
            case 4608: u = 5; break; case 256: u = 8; break; case 512: u = 9; break; case 1024: u = 10; break;
            case 2048: u = 11; break; case 4096: u = 12; break; case 8192: u = 13; break; case 16384: u = 14; break;
            case 32768: u = 15; break;
            default: bs_hint = u = (n <= 0x100) ? 6 : 7; break;
            }
            const uint32_t b2 = u << 4;
            const uint32_t sr = P.sample_rate;
            switch (sr) {
            case 88200: u = 1; break; case 176400: u = 2; break; case 192000: u = 3; break; case 8000: u = 4; break;
            case 16000: u = 5; break; case 22050: u = 6; break; case 24000: u = 7; break; case 32000: u = 8; break;
            case 44100: u = 9; break; case 48000: u = 10; break; case 96000: u = 11; break;
            default:
                if (sr <= 255000 && sr % 1000 == 0) sr_hint = u = 12;
                else if (sr <= 655350 && sr % 10 == 0) sr_hint = u = 14;
                else if (sr <= 0xffff) sr_hint = u = 13;
                else u = 0;
                break;
            }
            hb[hl++] = (uint8_t)(b2 | u);
            switch (ca) { case 0: u = P.channels - 1; break; case 1: u = 8; break; case 2: u = 9; break; default: u = 10; break; }
            const uint32_t b3 = u << 4;
            switch (P.bps) { case 8: u = 1; break; case 12: u = 2; break; case 16: u = 4; break; case 20: u = 5; break;
                             case 24: u = 6; break; case 32: u = 7; break; default: u = 0; break; }
            hb[hl++] = (uint8_t)(b3 | (u << 1));
            const uint32_t v = d.frame_number;
            if (v < 0x80) hb[hl++] = (uint8_t)v;
            else if (v < 0x800) { hb[hl++] = 0xC0 | (v >> 6); hb[hl++] = 0x80 | (v & 0x3F); }
            else if (v < 0x10000) { hb[hl++] = 0xE0 | (v >> 12); hb[hl++] = 0x80 | ((v >> 6) & 0x3F); hb[hl++] = 0x80 | (v & 0x3F); }
            else if (v < 0x200000) { hb[hl++] = 0xF0 | (v >> 18); hb[hl++] = 0x80 | ((v >> 12) & 0x3F); hb[hl++] = 0x80 | ((v >> 6) & 0x3F); hb[hl++] = 0x80 | (v & 0x3F); }
            else if (v < 0x4000000) { hb[hl++] = 0xF8 | (v >> 24); hb[hl++] = 0x80 | ((v >> 18) & 0x3F); hb[hl++] = 0x80 | ((v >> 12) & 0x3F); hb[hl++] = 0x80 | ((v >> 6) & 0x3F); hb[hl++] = 0x80 | (v & 0x3F); }
            else { hb[hl++] = 0xFC | (v >> 30); hb[hl++] = 0x80 | ((v >> 24) & 0x3F); hb[hl++] = 0x80 | ((v >> 18) & 0x3F); hb[hl++] = 0x80 | ((v >> 12) & 0x3F); hb[hl++] = 0x80 | ((v >> 6) & 0x3F); hb[hl++] = 0x80 | (v & 0x3F); }
            if (bs_hint == 6) hb[hl++] = (uint8_t)(n - 1);
            else if (bs_hint == 7) { hb[hl++] = (uint8_t)((n - 1) >> 8); hb[hl++] = (uint8_t)(n - 1); }
            if (sr_hint == 12) hb[hl++] = (uint8_t)(sr / 1000);
            else if (sr_hint == 13) { hb[hl++] = (uint8_t)(sr >> 8); hb[hl++] = (uint8_t)sr; }
            else if (sr_hint == 14) { hb[hl++] = (uint8_t)((sr / 10) >> 8); hb[hl++] = (uint8_t)(sr / 10); }
            uint32_t c8 = 0;
            for (uint32_t i = 0; i < hl; i++) {
                c8 ^= hb[i];
                for (int b = 0; b < 8; b++) c8 = (c8 & 0x80) ? (((c8 << 1) ^ 0x07) & 0xFF) : ((c8 << 1) & 0xFF);
            }
            hb[hl++] = (uint8_t)c8;
        }
        hl = rfl(hl);
        wave_lds_fence();
        const uint32_t v = (uint32_t)lane < hl ? hb[lane] : 0, b = (uint32_t)lane < hl ? 8 : 0;
        wave_lds_fence();
        fb_or(fb, (uint32_t)lane * 8, v, b);
        bitpos = hl * 8;
        wave_lds_fence();
    }
#pragma unroll 1
    for (uint32_t si = 0; si < (uint32_t)NCH; si++) {
        const uint32_t c = MS ? (si == 0 ? sub0 : sub1) : si;
        // select the decision of candidate c (wave-uniform)
        uint32_t type = d_type[0], order = d_order[0], sb = sbp[0], prec = d_prec[0], po = d_porder[0], method = d_method[0], kv = d_k[0];
        int shift = d_shift[0];
#pragma unroll
        for (int cc = 1; cc < NC; cc++)
            if (c == (uint32_t)cc) {
                type = d_type[cc]; order = d_order[cc]; sb = sbp[cc]; prec = d_prec[cc]; po = d_porder[cc]; method = d_method[cc];
                kv = d_k[cc]; shift = d_shift[cc];
            }
        const uint32_t mask = sb < 32 ? ((1u << sb) - 1) : 0xFFFFFFFFu;
        uint32_t hdr;
        switch (type) {
        case 0: hdr = 0x00; break;
        case 1: hdr = 0x02; break;
        case 2: hdr = 0x10 | (order << 1); break;
        default: hdr = 0x40 | ((order - 1) << 1); break;
        }
        auto cand = [&](int32_t l, int32_t r) -> int32_t {
            if (!MS) return c == 0 ? l : r;
            return c == 0 ? l : c == 1 ? r : c == 2 ? ((l + r) >> 1) : (l - r);
        };
        // ---- everything in front of the residual: lane 0 = subframe header byte, lanes 1..order = warm-up samples, then
        // precision/shift, coefficients, coding method + partition order; positions from one prefix sum
        {
            uint32_t pv = 0, pb = 0, val = 0, vb = 0;
            const bool pred = type >= 2;
            const uint32_t nw = type == 0 ? 1 : (pred ? order : 0);   // sample fields
            if (lane == 0) { pv = hdr; pb = 8; }
            else if ((uint32_t)lane <= nw) {
                const uint32_t g = (uint32_t)lane - 1;                 // sample index (inside segment 0: order <= MAXO <= seg)
                val = (uint32_t)cand(sL[g], (NCH == 2) ? sR[g] : 0) & mask; vb = sb;
            }
            else if (type == 3 && (uint32_t)lane == order + 1) { pv = prec - 1; pb = 4; val = (uint32_t)shift & 31; vb = 5; }
            else if (type == 3 && (uint32_t)lane <= 2 * order + 1) { val = (uint32_t)bestq[c * MAXO + (lane - order - 2)] & ((1u << prec) - 1); vb = prec; }
            else if (pred && (uint32_t)lane == (type == 3 ? 2 * order + 2 : order + 1)) { val = (method << 4) | po; vb = 6; }
            const uint32_t mine = pb + vb;
            const uint32_t incl = wave_scan_add(mine);
            const uint32_t total = rl(incl, 63);
            fb_reserve(fb, lane, bitpos, total);
            const uint32_t o = bitpos + incl - mine;
            fb_or(fb, o, pv, pb);
            fb_or(fb, o + pb, val, vb);
            bitpos += total;
            wave_lds_fence();
        }
        if (type == 0) continue;
        // ---- body: pass A = exact bit length of every lane's segment, prefix sum = its start, pass B = OR the codes in
        int32_t q[MAXO];
        if (type == 3) {
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = (int32_t)rfl((uint32_t)bestq[c * MAXO + j]);
        }
        else {
            const uint32_t g = type == 2 ? order : 0;
            const int32_t c0 = g == 0 ? 0 : (int32_t)g, c1 = g < 2 ? 0 : (g == 2 ? -1 : g == 3 ? -3 : -6);
            const int32_t c2 = g < 3 ? 0 : (g == 3 ? 1 : 4), c3 = g < 4 ? 0 : -1;
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : j == 3 ? c3 : 0;
            shift = 0;
        }
        const uint32_t plen = method ? 5 : 4;
        const uint32_t lpp = 64u >> po;                                  // lanes per partition
        const uint32_t kr = type >= 2 ? (uint32_t)__shfl((int)kv, (int)((uint32_t)lane & ~(lpp - 1))) : 0;   // this lane's Rice parameter
        const bool pstart = type >= 2 && ((uint32_t)lane % lpp) == 0;
        const uint32_t skip = (type >= 2 && lane == 0) ? order : 0;      // warm-up samples are not coded
        // one walk over the segment; EMIT = false: returns the bit length, EMIT = true: writes the codes from bit `p0` on
        // CC = candidate, VERB = verbatim subframe, EMIT: all compile-time, so the per-sample loop has no control flow
        // ALLF: every lane belongs to the group being emitted (the whole subframe fits the window: the common case)
        auto walk_t = [&](auto CC, auto VERB, auto EMIT, auto ATOM, auto ALLF, uint32_t p0, bool inrange_) __attribute__((always_inline)) -> uint32_t {
            constexpr int C_ = decltype(CC)::value;
            constexpr bool verb = decltype(VERB)::value, emit = decltype(EMIT)::value, atom = decltype(ATOM)::value;
            const bool inrange = decltype(ALLF)::value ? true : inrange_;
            int32_t h[MAXO];
#pragma unroll
            for (int j = 0; j < MAXO; j++) {
                int32_t x = 0;
                if (lane > 0) x = fcv<MS, C_>(rowL[(int)seg - 1 - j - (int)rstr], (NCH == 2) ? rowR[(int)seg - 1 - j - (int)rstr] : 0);
                h[(MAXO - 1 - j) % MAXO] = ACC64 ? fpack(x) : x;
            }
            uint32_t pos = p0, len = 0;
            // Emission without LDS atomics (they run at about one lane per clock on this hardware).  A lane's bits are
            // consecutive, so it keeps the word it is filling in a register (`cur`, window word `cw`) and stores it when
            // it moves on; the word it ends in is shared with the next lane and is merged after the walk.  The stores
            // are unconditional: a lane that has nothing to store writes to a scratch word of its own.
            // This needs every lane to span at least a word (no three lanes in one word): true for 32 or more samples
            // per lane; shorter segments (atom) OR their bits into the window with LDS atomics instead.
            LDS uint32_t *const dummy = misc + lane;             // misc[0..63]: header bytes earlier, free by now
            uint32_t cw = (p0 >> 5) - fb.wbase;
            uint32_t cur = (emit && !atom && inrange) ? fb.w[cw] : 0;
            auto put = [&](uint32_t at, uint32_t val, uint32_t vb) __attribute__((always_inline)) {
                if (atom) { fb_or(fb, inrange ? at : (fb.wbase << 5), inrange ? val : 0, vb); return; }
                const uint32_t rel = at - (fb.wbase << 5);
                const uint32_t wi = rel >> 5, sh = rel & 31;
                const u64 x = (u64)val << ((64 - sh - vb) & 63);
                const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
                const bool moved = inrange && wi != cw;
                *(moved ? fb.w + cw : dummy) = cur;
                cur = moved ? hi : (cur | hi);
                cw = wi;
                const bool spill = inrange && lo != 0;
                *(spill ? fb.w + cw : dummy) = cur;
                cur = spill ? lo : cur;
                cw += spill ? 1u : 0u;
            };
            if (pstart) {
                if (emit) put(pos, kr, plen);
                pos += plen; len += plen;
            }
            const uint32_t kmask = (1u << kr) - 1, kone = 1u << kr;
            // guard: the step may be one of the warm-up samples (the first `order` of the block, all in lane 0's first group)
            auto step = [&](int u, uint32_t s, bool guard) __attribute__((always_inline)) {
                const int32_t x = fcv<MS, C_>(rowL[s], (NCH == 2) ? rowR[s] : 0);
                uint32_t val, vb, lead;
                if (verb) { val = (uint32_t)x & mask; vb = sb; lead = 0; }
                else {
                    int32_t res;
                    if (!ACC64) res = x - (fir24<MAXO>(q, h, u) >> shift);
                    else res = (int32_t)((i64)x - (fir48<MAXO>(q, h, u) >> shift));
                    h[u] = ACC64 ? fpack(x) : x;
                    const uint32_t uu = ((uint32_t)res << 1) ^ (uint32_t)(res >> 31);
                    lead = uu >> kr;
                    val = kone | (uu & kmask);
                    vb = kr + 1;
                }
                if (guard) {
                    const bool coded = s >= skip;
                    if (emit) put(coded ? pos + lead : pos, coded ? val : 0, coded ? vb : 0);     // warm-up samples: nothing, in place
                    const uint32_t cl_ = coded ? lead + vb : 0;
                    pos += cl_; len += cl_;
                }
                else {
                    if (emit) put(pos + lead, val, vb);
                    pos += lead + vb; len += lead + vb;
                }
            };
            uint32_t s0 = 0;
            if (seg >= (uint32_t)MAXO) {
#pragma unroll
                for (int u = 0; u < MAXO; u++) step(u, (uint32_t)u, true);
                s0 = MAXO;
            }
#pragma unroll 1
            for (; s0 + MAXO <= seg; s0 += MAXO) {
#pragma unroll
                for (int u = 0; u < MAXO; u++) step(u, s0 + u, false);
            }
#pragma unroll
            for (int u = 0; u < MAXO; u++) if (s0 + u < seg) step(u, s0 + u, s0 == 0);
            if (emit && !atom) {
                // the last word of every lane: no other lane of the group ends in it (each lane spans at least a word)
                wave_lds_fence();
                if (inrange) fb.w[cw] |= cur;
                wave_lds_fence();
            }
            return len;
        };
        auto walk_c = [&](auto CC, bool emit, uint32_t p0, bool inrange, bool all) __attribute__((always_inline)) -> uint32_t {
            typedef std::integral_constant<bool, true> T;
            typedef std::integral_constant<bool, false> F;
            if (!emit) return type == 1 ? walk_t(CC, T(), F(), F(), F(), p0, inrange) : walk_t(CC, F(), F(), F(), F(), p0, inrange);
            if (seg < 32) return type == 1 ? walk_t(CC, T(), T(), T(), F(), p0, inrange) : walk_t(CC, F(), T(), T(), F(), p0, inrange);
            if (all) return type == 1 ? walk_t(CC, T(), T(), F(), T(), p0, true) : walk_t(CC, F(), T(), F(), T(), p0, true);
            return type == 1 ? walk_t(CC, T(), T(), F(), F(), p0, inrange) : walk_t(CC, F(), T(), F(), F(), p0, inrange);
        };
        auto walk = [&](bool emit, uint32_t p0, bool inrange, bool all) __attribute__((always_inline)) -> uint32_t {
            if (NC == 1 || c == 0) return walk_c(std::integral_constant<int, 0>(), emit, p0, inrange, all);
            if (NC == 2 || c == 1) return walk_c(std::integral_constant<int, (NC > 1 ? 1 : 0)>(), emit, p0, inrange, all);
            if (c == 2) return walk_c(std::integral_constant<int, (NC > 2 ? 2 : 0)>(), emit, p0, inrange, all);
            return walk_c(std::integral_constant<int, (NC > 3 ? 3 : 0)>(), emit, p0, inrange, all);
        };
        FG_TACC(0);
        const uint32_t mylen = walk(false, 0, false, false);
        FG_TACC(1);
        if (__any(mylen > (1u << 24))) { fb.err |= FG_ERR_REDO; break; }       // absurd code lengths: the generic kernel copes
        const uint32_t incl = wave_scan_add(mylen);
        const uint32_t mystart = bitpos + incl - mylen, myend = bitpos + incl;
        const uint32_t subend = bitpos + rl(incl, 63);
        // lanes are emitted in ascending groups that fit the window
        uint32_t a = 0;
        bool failed = false;
#pragma unroll 1
        while (a < 64) {
            fb_flush(fb, lane, rl(mystart, (int)a));
            const uint32_t cap = (fb.wbase << 5) + 32u * FGS_FBW - 64u;
            const uint64_t fits = __ballot((uint32_t)lane >= a && myend <= cap);
            // lanes a .. b-1 fit (ends are ascending, so the fitting lanes form a prefix of a..63)
            const uint64_t shifted = fits >> a;
            const uint32_t cnt = (~shifted) ? (uint32_t)__builtin_ctzll(~shifted) : 64u - a;
            if (cnt == 0) { failed = true; break; }
            const uint32_t b = a + cnt;
            FG_TACC(0);
            (void)walk(true, mystart, (uint32_t)lane >= a && (uint32_t)lane < b, a == 0 && b == 64);
            wave_lds_fence();
            FG_TACC(2);
            a = b;
        }
        if (failed) { fb.err |= FG_ERR_REDO; break; }
        bitpos = subend;
    }
    FG_TACC(0);
    FG_STAMP(8);
    if (mydbg && lane == 0) { mydbg->t[10] = tacc[0]; mydbg->t[11] = tacc[1]; mydbg->t[12] = tacc[2]; }
    if (fb.err & FG_ERR_REDO) {
        if (lane == 0) {
            FgBlockResult *r = &results[d.out_slot];
            r->bytes = 0; r->ca = 0; r->err = FG_ERR_REDO; r->reserved = 3;
        }
        return;
    }
    // ---- zero-pad to a byte, CRC-16 over the whole frame, append
    if (bitpos & 7) bitpos += 8 - (bitpos & 7);
    fb_flush(fb, lane, bitpos);
    {
        const uint32_t nbytes = bitpos >> 3;
        const uint32_t W = nbytes >> 2, tail = nbytes & 3;
        // lane l holds the CRC state of the words l, l+64, ...; its last word is dist = (W - 1 - l) mod 64 words from the end
        uint32_t s = 0;
        if ((uint32_t)lane < W) s = gf16_mul(fb.crc, misc[64 + ((W - 1 - (uint32_t)lane) & 63)]);
        uint32_t crc = wave_xor32(s);
        if (tail) {
            const uint32_t wv = rfl(fbw[0]);
            for (uint32_t b = 0; b < tail; b++) crc = ((crc << 8) & 0xFFFF) ^ crct[((crc >> 8) ^ (wv >> (24 - 8 * b))) & 0xFF];
        }
        if (lane == 0) fb_or(fb, bitpos, crc, 16);
        bitpos += 16;
        wave_lds_fence();
        fb_flush(fb, lane, bitpos);
        if ((bitpos & 31) && lane == 0) {
            if (fb.wbase < fb.slot_words) fb.outw[fb.wbase] = __builtin_bswap32(fbw[0]);
        }
        if ((bitpos & 31) && fb.wbase >= fb.slot_words) fb.err |= FG_ERR_SLOT;
    }
    FG_STAMP(9);
    if (lane == 0) {
        FgBlockResult *r = &results[d.out_slot];
        r->bytes = bitpos >> 3; r->ca = ca; r->err = err | fb.err; r->reserved = 1;
#pragma unroll
        for (int c = 0; c < 4; c++) r->best_bits[c] = c < NC ? best[c < NC ? c : 0] : 0;
    }
#undef FG_STAMP
#undef FG_SADDR
}

}  // namespace
