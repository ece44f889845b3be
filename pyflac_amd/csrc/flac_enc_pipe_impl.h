// flac_enc_pipe_impl.h -- the de-fused FLAC frame encoder for gfx950: five kernels per launch instead of one.
//
// Round 1's single kernel (flac_enc_fast_impl.h) kept a block in one wavefront from PCM to frame.  It needed 240 VGPRs and
// 22 KB of LDS per block, so 1.75 waves per SIMD were resident and no stage could hide its latency (dependent VALU chains,
// the 4096-step fp64 autocorrelation chain, LDS round trips).  The stages want different mappings and different register
// budgets, and HBM bandwidth is nowhere near a limit here (the encoder moves ~0.3 GB per launch), so the stages are
// separate kernels that hand small records through HBM and re-read the PCM (it stays in the Infinity Cache):
//
//   K2 fg_pipe_autoc_kernel     wave = block.  Windowed signal straight from HBM, staged per 128-sample chunk as doubles;
//                               libFLAC's order-preserving fp64 chains run on the matrix core: one
//                               v_mfma_f64_4x4x4_4b_f64 adds four steps to the chains of 16 lags of all four candidates
//                               (its k-sum is a chain of fused multiply-adds in ascending k, bit-equal to v_fma_f64).
//                               5.2 KB of LDS and <= 64 VGPRs per wave: 7-8 waves per SIMD.  Also ORs the samples: wasted
//                               bits per candidate.  The autocorrelation of the shifted signal is the autocorrelation
//                               of the unshifted one times 2^-2w, exactly (power-of-two scaling commutes with every
//                               rounding on the way), so the chain never waits for the wasted-bits result.
//   K3 fg_pipe_levinson_kernel  lane = (block, candidate, window): Levinson-Durbin, order guess, quantiser.  The round-1
//                               kernel ran this on 4 of 64 lanes.
//   K4 fg_pipe_eval_kernel      workgroup = block, wave = candidate (L, R, M, S).  Samples staged once in LDS (lane =
//                               segment rows); fixed-predictor error sums, constant detection, FIR residual of the LPC
//                               candidates (history in VGPRs, coefficients in SGPRs), Rice partition search.  One
//                               candidate per wave keeps it under 64 VGPRs: 8 blocks = 32 waves per CU.
//   K5 fg_pipe_pack_kernel      workgroup = block, wave = (subframe, half).  Each wave Rice-codes its 64 segments into a
//                               private word-aligned chunk (exact lengths, prefix sum, register-assembled words -- the
//                               round-1 packer) with no cross-wave dependency.
//   K6 fg_pipe_assemble_kernel  wave = frame.  Concatenates the chunks at bit granularity (funnel shifts), pads, computes
//                               the CRC-16 and writes the frame at its final byte offset in the output stream (this
//                               replaces the slot -> stream compaction copy of round 1).
//
// Bytes are identical to the round-1 kernels and to libFLAC 1.4.3 (tests/test_gpu_encode.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fg_dev.h"
#include "fg_types.h"

#define FG_LN2 0.69314718055994530942
#define LDS __attribute__((address_space(3)))
#define FGI __device__ __forceinline__

// (P.debug values 101..103 switch parts of the autocorrelation kernel off: timing experiments of round 3, compiled into tuning builds only)
#ifdef FG_TUNING
#define FGX_DBG(v) (P.debug == (v))
#else
#define FGX_DBG(v) false
#endif
#ifndef FGX_AUTOC_RD2
#define FGX_AUTOC_RD2 1
#endif
#ifndef FGX_MERGE
// evaluation (<= 16 bit, regular geometry): fixed-predictor sums and the first LPC vector's residual in ONE walk over the samples (the
// candidate's samples formed once instead of twice: 3 instructions a sample fewer).  Measured slower on the MI355X, 124.4 -> 128.1 us
// for the kernel on its own (gpurun_exp m0c0 / m1c0, one group): the merged loop spills four registers at the kernel's 64, and the
// two short loops it replaces interleave better.  Off; kept for the A/B.
#define FGX_MERGE 0
#endif
#ifndef FGX_CRCTAB
#define FGX_CRCTAB 1                     // direct packing: CRC-16 of the frame through look-up tables in LDS (0: the closed form)
#endif
#define FGP_F64P 1                       // packing: the same (with the 168 registers that three workgroups per CU leave: 24-bit level 8 0.28 -> 0.25 ms)
#define FGP_F64 1                        // evaluation: residuals of 17..25-bit samples through fp64 FMAs (pfir_f64n; 0 = pfir48)
#define FGP_DH 16                        // autocorrelation: history doubles kept in front of each chunk (>= max lag + 1)
#define FGP_CK 128                       // autocorrelation: chunk length
// doubles per candidate row: 304 words, i.e. 48 banks (of 64) from row to row -- the four candidate rows that share a
// ds_read_b64 lane group read 7 consecutive doubles each (d[t + k - i], 14 banks) and never meet on a bank.
// 5.2 KB of LDS per wave keeps 28+ waves on a CU: the 7032 blocks of the headline stream are resident in one round.
#define FGP_SLK 8                        // zeros behind a chunk: the chains run up to 7 steps past the end of the signal
#define FGP_CSTR (FGP_DH + FGP_CK + FGP_SLK)

using namespace fgdev;

namespace {

template <bool ACC64> struct PipeTypes {
    typedef typename std::conditional<ACC64, u64, uint32_t>::type sum_t;
    typedef typename std::conditional<ACC64, int32_t, int16_t>::type samp_t;   // <= 16 bit input is staged as int16
    static constexpr uint32_t PADE = 2;   // elements of skew between rows: odd word stride, no bank conflicts
};

template <bool MS, int C> FGI int32_t pcv(int32_t L, int32_t R)
{
    if (!MS) return C == 0 ? L : R;
    if (C == 0) return L;
    if (C == 1) return R;
    if (C == 2) return (L + R) >> 1;
    return L - R;
}

// a pointer the caller knows to be the same in every lane, made so for the compiler too (it then lives in scalar registers and a
// load takes it as the base beside a 32-bit lane offset)
template <typename T> FGI const T *uniform_ptr(const T *p)
{
    const u64 v = (u64)(uintptr_t)p;
    const uint32_t lo = rfl((uint32_t)v), hi = rfl((uint32_t)(v >> 32));
    return (const T *)(uintptr_t)(((u64)hi << 32) | lo);
}

FGI uint32_t pabs32(int32_t v) { return (uint32_t)(v < 0 ? -v : v); }
// |a - b| + c on unsigned operands in one instruction
FGI uint32_t psad(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// sum_j q[j] * h[(u - 1 - j) mod MAXO]: one v_mad_i32_i24 per tap, the coefficient from an SGPR (one candidate per wave, so
// the MAXO coefficients are wave-uniform and fit the scalar file).  History slot of sample s is s mod MAXO.
template <int MAXO> FGI int32_t pfir24(const int32_t (&q)[MAXO], const int32_t (&h)[MAXO], int u)
{
    // (the first tap is a multiply into a register of its own: a sum that starts as a zero costs a v_mov_b32 a sample)
    int32_t sm;
#define FG_H(j) h[(u - 1 - (j) + 2 * MAXO) % MAXO]
    if (MAXO == 8) {
        asm("v_mul_i32_i24 %0, %1, %9\n\tv_mad_i32_i24 %0, %2, %10, %0\n\tv_mad_i32_i24 %0, %3, %11, %0\n\t"
            "v_mad_i32_i24 %0, %4, %12, %0\n\tv_mad_i32_i24 %0, %5, %13, %0\n\tv_mad_i32_i24 %0, %6, %14, %0\n\t"
            "v_mad_i32_i24 %0, %7, %15, %0\n\tv_mad_i32_i24 %0, %8, %16, %0"
            : "=&v"(sm)
            : "s"(q[7 % MAXO]), "s"(q[6 % MAXO]), "s"(q[5 % MAXO]), "s"(q[4 % MAXO]), "s"(q[3 % MAXO]), "s"(q[2 % MAXO]), "s"(q[1 % MAXO]), "s"(q[0]),
              "v"(FG_H(7)), "v"(FG_H(6)), "v"(FG_H(5)), "v"(FG_H(4)), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
    }
    else {
        asm("v_mul_i32_i24 %0, %1, %7\n\tv_mad_i32_i24 %0, %2, %8, %0\n\tv_mad_i32_i24 %0, %3, %9, %0\n\t"
            "v_mad_i32_i24 %0, %4, %10, %0\n\tv_mad_i32_i24 %0, %5, %11, %0\n\tv_mad_i32_i24 %0, %6, %12, %0"
            : "=&v"(sm)
            : "s"(q[11 % MAXO]), "s"(q[10 % MAXO]), "s"(q[9 % MAXO]), "s"(q[8 % MAXO]), "s"(q[7 % MAXO]), "s"(q[6 % MAXO]),
              "v"(FG_H(11)), "v"(FG_H(10)), "v"(FG_H(9)), "v"(FG_H(8)), "v"(FG_H(7)), "v"(FG_H(6)));
        asm("v_mad_i32_i24 %0, %1, %7, %0\n\tv_mad_i32_i24 %0, %2, %8, %0\n\tv_mad_i32_i24 %0, %3, %9, %0\n\t"
            "v_mad_i32_i24 %0, %4, %10, %0\n\tv_mad_i32_i24 %0, %5, %11, %0\n\tv_mad_i32_i24 %0, %6, %12, %0"
            : "+v"(sm)
            : "s"(q[5 % MAXO]), "s"(q[4 % MAXO]), "s"(q[3 % MAXO]), "s"(q[2 % MAXO]), "s"(q[1 % MAXO]), "s"(q[0]),
              "v"(FG_H(5)), "v"(FG_H(4)), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
    }
#undef FG_H
    return sm;
}
// 17..25-bit samples: the exact 64-bit sum from two 16 x 16-bit MAD chains over a history packed as (x >> 12, x & 0xFFF)
// (flac_enc_fast_impl.h fir48 has the derivation); coefficients from SGPRs.
FGI int32_t ppack(int32_t x) { return (int32_t)(((uint32_t)(x >> 12) << 16) | ((uint32_t)x & 0xFFFu)); }
template <int MAXO> FGI i64 pfir48(const int32_t (&q)[MAXO], const int32_t (&h)[MAXO], int u)
{
    int32_t sl = 0, sh = 0;
#define FG_H(j) h[(u - 1 - (j) + 2 * MAXO) % MAXO]
#define FG_M6(OPS) "v_mad_i32_i16 %0, %1, %7, %0 " OPS "\n\tv_mad_i32_i16 %0, %2, %8, %0 " OPS "\n\tv_mad_i32_i16 %0, %3, %9, %0 " OPS "\n\t" \
                   "v_mad_i32_i16 %0, %4, %10, %0 " OPS "\n\tv_mad_i32_i16 %0, %5, %11, %0 " OPS "\n\tv_mad_i32_i16 %0, %6, %12, %0 " OPS
#define FG_M4(OPS) "v_mad_i32_i16 %0, %1, %5, %0 " OPS "\n\tv_mad_i32_i16 %0, %2, %6, %0 " OPS "\n\tv_mad_i32_i16 %0, %3, %7, %0 " OPS "\n\t" \
                   "v_mad_i32_i16 %0, %4, %8, %0 " OPS
    if (MAXO == 8) {
        asm(FG_M4("op_sel:[0,0,0,0]") : "+v"(sl) : "s"(q[7 % MAXO]), "s"(q[6 % MAXO]), "s"(q[5 % MAXO]), "s"(q[4 % MAXO]), "v"(FG_H(7)), "v"(FG_H(6)), "v"(FG_H(5)), "v"(FG_H(4)));
        asm(FG_M4("op_sel:[0,0,0,0]") : "+v"(sl) : "s"(q[3 % MAXO]), "s"(q[2 % MAXO]), "s"(q[1 % MAXO]), "s"(q[0]), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
        asm(FG_M4("op_sel:[0,1,0,0]") : "+v"(sh) : "s"(q[7 % MAXO]), "s"(q[6 % MAXO]), "s"(q[5 % MAXO]), "s"(q[4 % MAXO]), "v"(FG_H(7)), "v"(FG_H(6)), "v"(FG_H(5)), "v"(FG_H(4)));
        asm(FG_M4("op_sel:[0,1,0,0]") : "+v"(sh) : "s"(q[3 % MAXO]), "s"(q[2 % MAXO]), "s"(q[1 % MAXO]), "s"(q[0]), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
    }
    else {
        asm(FG_M6("op_sel:[0,0,0,0]") : "+v"(sl) : "s"(q[11 % MAXO]), "s"(q[10 % MAXO]), "s"(q[9 % MAXO]), "s"(q[8 % MAXO]), "s"(q[7 % MAXO]), "s"(q[6 % MAXO]),
            "v"(FG_H(11)), "v"(FG_H(10)), "v"(FG_H(9)), "v"(FG_H(8)), "v"(FG_H(7)), "v"(FG_H(6)));
        asm(FG_M6("op_sel:[0,0,0,0]") : "+v"(sl) : "s"(q[5 % MAXO]), "s"(q[4 % MAXO]), "s"(q[3 % MAXO]), "s"(q[2 % MAXO]), "s"(q[1 % MAXO]), "s"(q[0]),
            "v"(FG_H(5)), "v"(FG_H(4)), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
        asm(FG_M6("op_sel:[0,1,0,0]") : "+v"(sh) : "s"(q[11 % MAXO]), "s"(q[10 % MAXO]), "s"(q[9 % MAXO]), "s"(q[8 % MAXO]), "s"(q[7 % MAXO]), "s"(q[6 % MAXO]),
            "v"(FG_H(11)), "v"(FG_H(10)), "v"(FG_H(9)), "v"(FG_H(8)), "v"(FG_H(7)), "v"(FG_H(6)));
        asm(FG_M6("op_sel:[0,1,0,0]") : "+v"(sh) : "s"(q[5 % MAXO]), "s"(q[4 % MAXO]), "s"(q[3 % MAXO]), "s"(q[2 % MAXO]), "s"(q[1 % MAXO]), "s"(q[0]),
            "v"(FG_H(5)), "v"(FG_H(4)), "v"(FG_H(3)), "v"(FG_H(2)), "v"(FG_H(1)), "v"(FG_H(0)));
    }
#undef FG_M6
#undef FG_M4
#undef FG_H
    return (i64)(((u64)(i64)sh) << 12) + (i64)sl;
}

// The same sum in fp64: samples below 2^25 and coefficients below 2^15 give products below 2^40 and sums of twelve below
// 2^44 -- every fused multiply-add is exact, one instruction per tap (v_fma_f64 runs at the rate of the 32-bit integer
// operations on this chip; the two 16-bit chains above take two).  Two chains of half the taps each, for the latency.
// The sum of the taps in fp64 on top of `init`, one chain or two (FGP_F64_CH): the evaluation and the packer start it at -x with the
// coefficients scaled by 2^-shift -- x - floor(p 2^-shift) = -floor(p 2^-shift - x), x being an integer --, which saves the
// scaling and the subtraction of every sample.  Everything is a multiple of 2^-shift below 2^44: every step is exact, in any order.
#ifndef FGP_F64_CH
#define FGP_F64_CH 1
#endif
template <int MAXO> FGI double pfir_f64n(const double (&q)[MAXO], const double (&h)[MAXO], int u, double init)
{
#define FG_H(j) h[(u - 1 - (j) + 2 * MAXO) % MAXO]
    if (FGP_F64_CH == 1) {
        double a = init;
#pragma unroll
        for (int j = MAXO - 1; j >= 0; j--) a = __builtin_fma(q[j], FG_H(j), a);
        return a;
    }
    double a = init, b = 0.0;
#pragma unroll
    for (int j = MAXO - 1; j >= MAXO / 2; j--) a = __builtin_fma(q[j], FG_H(j), a);
#pragma unroll
    for (int j = MAXO / 2 - 1; j >= 0; j--) b = __builtin_fma(q[j], FG_H(j), b);
#undef FG_H
    return a + b;
}

FGI double p_ebps(double e, double scale)
{
    if (e > 0.0) {
        const double bb = 0.5 * log(scale * e) / FG_LN2;
        return bb >= 0.0 ? bb : 0.0;
    }
    else if (e < 0.0) return 1e32;
    return 0.0;
}

// ================================================================================================ K1: start of a launch
// The device stamp of the call and the counters the later kernels add to: near-tie guard of the LPC order guess (count 0,
// smallest margin +infinity) and the OR of the error flags of the pipeline's blocks.  A kernel of its own since the groups of
// a launch run on several streams (FgPipeLaunch.ngroups): everything that touches the counters is ordered behind it.
__global__ void fg_pipe_begin_kernel(FgPipeBufs B)
{
    if (B.stamp) B.stamp[0] = wall_clock64();
    if (B.guard) { B.guard[0] = 0ull; B.guard[1] = 0x7FF0000000000000ull; B.guard[2] = 0ull; }
}

// ================================================================================================ K2: autocorrelation + OR
// (<= 80 SGPRs: above that the scalar file allows only 7 waves per SIMD, 7168 on the chip -- the 7032 blocks of the headline
// stream would then only fit with a perfectly even spread over the CUs, and the stragglers would double the kernel's time)
// Four independent waves (blocks) per workgroup: a CU holds at most 16 workgroups, so one-wave workgroups would leave half
// of its 32 wave slots empty.  No barriers: the waves share nothing but the workgroup's LDS allocation.
#define FGP_AWPB 4
template <bool MS, int NCH, int MAXO>
__global__ void __launch_bounds__(64 * FGP_AWPB, 8) __attribute__((amdgpu_num_sgpr(72)))
fg_pipe_autoc_kernel(const void *pcm, const FgBlockDesc *descs, const float *windows, FgEncParams P, FgPipeBufs B, FgDebugRec *dbg,
                     uint32_t nblocks, uint32_t lds_per_wave, uint32_t bi0)
{
    constexpr int NC = MS ? 4 : NCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t wv = rfl(threadIdx.x >> 6);
    const uint32_t bi = (bi0 & 0x3FFFFFFFu) + blockIdx.x * FGP_AWPB + wv;           // (blocks [bi0, nblocks) of the list: one group of the launch)
    // (bit 31 of bi0: this is the only group of the launch -- the first workgroup does what fg_pipe_begin_kernel does otherwise, and a
    // call of one block, StreamEncoder.process with libFLAC's timing, has one kernel less to wait for.  Bit 30: the first group of
    // several, the counters already reset by the previous call's signal kernel: only the stamp is taken)
    if ((bi0 >> 30) && blockIdx.x == 0 && threadIdx.x == 0) {
        if (B.stamp) B.stamp[0] = wall_clock64();
        if ((bi0 >> 31) && B.guard) { B.guard[0] = 0ull; B.guard[1] = 0x7FF0000000000000ull; B.guard[2] = 0ull; }
    }
    if (bi >= nblocks) return;
    const FgBlockDesc d = descs[bi];
    const int lane = threadIdx.x & 63;
    const uint32_t n = d.n;
    LDS double *dbuf = (LDS double *)((LDS unsigned char *)smem + wv * lds_per_wave);   // NC rows of FGP_CSTR doubles
    LDS double *autoc = dbuf + NC * FGP_CSTR;                   // [NC][nvec][MAXO + 1]
    LDS uint32_t *wl = (LDS uint32_t *)(autoc + NC * P.nvec * (MAXO + 1));   // wasted bits per candidate
    // (uniform bases, made so explicitly: the window of this block length and the block's first sample -- see the chunk fetch below)
    const float *const window = uniform_ptr(windows + d.win_off);
    const unsigned char *const pcmb = uniform_ptr((const unsigned char *)pcm + d.pcm_off * (NCH == 2 ? (P.pcm_i16 ? 4u : 8u) : (P.pcm_i16 ? 2u : 4u)));
    FgDebugRec *mydbg = dbg ? dbg + d.out_slot : nullptr;
    uint32_t orv[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) orv[c] = 0;
    // sample i of the block (both channels)
    auto ldsamp = [&](uint32_t i, int32_t &L, int32_t &R) __attribute__((always_inline)) {
        if (NCH == 2) {
            if (P.pcm_i16) { const short2 v = ((const short2 *)pcmb)[i]; L = v.x; R = v.y; }
            else { const int2 v = ((const int2 *)pcmb)[i]; L = v.x; R = v.y; }
        }
        else {
            // (one channel of an interleaved stream of more channels: the block is a view with a stride, see FgBlockDesc)
            const uint32_t at = i * (d.reserved ? d.reserved : 1u);
            if (P.pcm_i16) L = ((const int16_t *)pcmb)[at];
            else L = ((const int32_t *)pcmb)[at];
            R = 0;
        }
    };
    // The same sample as it lies in memory, no arithmetic on it: what the chunk loop fetches a chunk ahead.  (ldsamp's sign
    // extension of a 16-bit pair sits right behind the load, and the compiler's s_waitcnt vmcnt(0) with it -- in front of the
    // chain the load was meant to travel under: every chunk then waited out two HBM latencies.)  unraw() makes L and R of it at
    // the point of use; zeros stay zeros.
    auto ldraw = [&](uint32_t i, int32_t &a, int32_t &b) __attribute__((always_inline)) {
        if (NCH == 2 && P.pcm_i16) { a = ((const int32_t *)pcmb)[i]; b = 0; }
        else ldsamp(i, a, b);
    };
    auto unraw = [&](int32_t &L, int32_t &R) __attribute__((always_inline)) {
        if (NCH == 2 && P.pcm_i16) { R = L >> 16; L = (int32_t)(int16_t)L; }
    };
    // candidate c of a sample as the float libFLAC windows (FLAC__lpc_window_data: the integer converted to float, one rounding)
    // and the bits its wasted-bits count looks at.  32-bit input: mid needs a 33-bit sum and the side channel IS 33 bits wide
    // (libFLAC keeps it in 64 bits and windows it with FLAC__lpc_window_data_wide); the low 32 bits decide the trailing zeros.
    auto cval = [&](int c, int32_t L, int32_t R, float &fx, uint32_t &ob) __attribute__((always_inline)) {
        if (MS && P.bps == 32 && c == 2) { const int32_t m_ = (int32_t)(((i64)L + (i64)R) >> 1); fx = (float)m_; ob = (uint32_t)m_; }
        else if (MS && P.bps == 32 && c == 3) { fx = (float)((double)L - (double)R); ob = (uint32_t)L - (uint32_t)R; }
        else {
            const int32_t x = c == 0 ? pcv<MS, 0>(L, R) : c == 1 ? pcv<MS, 1>(L, R) : c == 2 ? pcv<MS, 2>(L, R) : pcv<MS, 3>(L, R);
            fx = (float)x; ob = (uint32_t)x;
        }
    };
    uint32_t nv = 0;
    const uint32_t mo = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
    if (mo == 0) {
        // no LPC at this level: only the OR of the samples is needed
        for (uint32_t i = lane; i < n; i += 64) {
            int32_t L, R;
            ldsamp(i, L, R);
#pragma unroll
            for (int c = 0; c < NC; c++) { float fx; uint32_t ob; cval(c, L, R, fx, ob); orv[c] |= ob; }
        }
    }
    else {
        // (matrix-core chains, see below: lane -> candidate row; lanes of candidates this shape does not have read row 0)
        const uint32_t mrow = ((lane >> 2) & 3) < (uint32_t)NC ? ((lane >> 2) & 3) : 0;
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
                for (uint32_t j = lane; j < NC * FGP_DH; j += 64) dbuf[(j / FGP_DH) * FGP_CSTR + (j % FGP_DH)] = 0.0;
                for (uint32_t j = lane; j < NC * FGP_SLK; j += 64) dbuf[(j / FGP_SLK) * FGP_CSTR + FGP_DH + FGP_CK + (j % FGP_SLK)] = 0.0;
                wave_lds_fence();
                // window value and samples of the next chunk travel while the chain of the current one runs
                float wv[FGP_CK / 64];
                int32_t xl[FGP_CK / 64], xr[FGP_CK / 64];
                auto fetch = [&](uint32_t k0) __attribute__((always_inline)) {
                    if (part == 0 && k0 + FGP_CK <= vec_len) {
                        // (the whole block under one window, a full chunk: nearly every call)
                        // Round 5: the chunk's addresses are a uniform base (scalar registers, made so explicitly) plus the lane's
                        // offset.  Left to itself the compiler kept a 64-bit pointer per lane, load and loop variant alive across
                        // the whole kernel -- a dozen vector registers beside the twenty the chain owns -- and spilled 100 bytes a
                        // lane to scratch: 45 MB written and as much read back per launch of the headline stream.
                        if (NCH == 2) {
                            const float *wk = uniform_ptr(window + k0);
                            if (P.pcm_i16) {
                                const int32_t *pk = uniform_ptr((const int32_t *)pcmb + k0);
#pragma unroll
                                for (int u = 0; u < FGP_CK / 64; u++) { wv[u] = wk[u * 64 + lane]; xl[u] = pk[u * 64 + lane]; xr[u] = 0; }
                            }
                            else {
                                const int2 *pk = uniform_ptr((const int2 *)pcmb + k0);
#pragma unroll
                                for (int u = 0; u < FGP_CK / 64; u++) { wv[u] = wk[u * 64 + lane]; const int2 v = pk[u * 64 + lane]; xl[u] = v.x; xr[u] = v.y; }
                            }
                            return;
                        }
#pragma unroll
                        for (int u = 0; u < FGP_CK / 64; u++) {
                            const uint32_t i = k0 + u * 64 + lane;
                            wv[u] = window[i];
                            ldraw(i, xl[u], xr[u]);
                        }
                        return;
                    }
#pragma unroll
                    for (int u = 0; u < FGP_CK / 64; u++) {
                        const uint32_t i = k0 + u * 64 + lane;
                        float w = 0.0f;
                        int32_t L = 0, R = 0;
                        if (i < vec_len) {
                            uint32_t s_ = 0;
                            bool any = true;
                            if (part == 0) { w = window[i]; s_ = i; }
                            else if (i < part) { w = window[i]; s_ = sh + i; }
                            else if (i < 2 * part) { w = window[n - 2 * part + i]; s_ = sh + i; }
                            else any = false;
                            if (any) ldraw(s_, L, R);
                        }
                        wv[u] = w; xl[u] = L; xr[u] = R;
                    }
                };
                fetch(0);
                for (uint32_t k0 = 0; k0 < vec_len; k0 += FGP_CK) {
                    const uint32_t kn = (vec_len - k0) < FGP_CK ? (vec_len - k0) : FGP_CK;
                    if (part == 0 && P.bps < 32) {
#pragma unroll
                        for (int u = 0; u < FGP_CK / 64; u++) {
                            const uint32_t j = u * 64 + lane;
                            if (!FGX_DBG(102)) {       // (behind the end of the signal the fetch gave zeros: the chains run on a little)
                                int32_t L = xl[u], R = xr[u];
                                unraw(L, R);
#pragma unroll
                                for (int c = 0; c < NC; c++) {
                                    const int32_t x = c == 0 ? pcv<MS, 0>(L, R) : c == 1 ? pcv<MS, 1>(L, R) : c == 2 ? pcv<MS, 2>(L, R) : pcv<MS, 3>(L, R);
                                    orv[c] |= (uint32_t)x;       // (windows after the first see samples the first has seen: the OR stands)
                                    dbuf[c * FGP_CSTR + FGP_DH + j] = (double)((float)x * wv[u]);
                                }
                            }
                        }
                    }
                    else {
#pragma unroll
                        for (int u = 0; u < FGP_CK / 64; u++) {
                            const uint32_t j = u * 64 + lane;
                            if (!FGX_DBG(102)) {
                                int32_t L = xl[u], R = xr[u];
                                unraw(L, R);
                                const bool zero = part != 0 && (k0 + j) >= 2 * part;
#pragma unroll
                                for (int c = 0; c < NC; c++) {
                                    float fx;
                                    uint32_t ob;
                                    cval(c, L, R, fx, ob);
                                    if (nv == 0) orv[c] |= ob;
                                    const float dd = zero ? 0.0f : fx * wv[u];
                                    dbuf[c * FGP_CSTR + FGP_DH + j] = (double)dd;
                                }
                            }
                        }
                    }
                    if (k0 + FGP_CK < vec_len) fetch(k0 + FGP_CK);
                    wave_lds_fence();
                    if (!FGX_DBG(101)) {
                        // The chains on the matrix core.  v_mfma_f64_4x4x4_4b_f64 computes, for four independent blocks b,
                        // D[i][j] += sum_k A[i][k] * B[k][j] with one fused multiply-add per k, in ascending k, each rounded like
                        // v_fma_f64 (tools/ubench/mfma64.hip: 262144 random cases bit-equal to that chain and to no other
                        // order).  Lanes: A and B in lane k*16 + b*4 + x (x = i resp. j), D in lane i*16 + b*4 + j.  With
                        //     A[i][k] = d[t + k - i]        B[k][j] = d[t + k - 4j]
                        // one instruction adds steps t .. t+3, in order, to the chains of lags 4j - i = -3 .. 12 of the four
                        // candidates b: 256 of libFLAC's fused multiply-adds, in libFLAC's order per lag (row i runs i steps
                        // behind; zeros in front of the signal and behind it add nothing; lags -1 .. -3 repeat 1 .. 3 and are
                        // dropped).  A comes from LDS (one ds_read_b64 per instruction).  B of the next step group is B of this
                        // one moved up a lane inside each quad, with the A just read entering at j = 0 (the same d[t + k]):
                        // two v_cndmask_b32_dpp.
                        const uint32_t kend = (k0 + kn < vec_len) ? FGP_CK : ((kn + 3) & ~3u) + 4;
                        const LDS double *pa = dbuf + mrow * FGP_CSTR + FGP_DH + (lane >> 4) - (lane & 3);
                        // One asm block per chunk (fixed registers v40..v59: A of two groups of four instructions in flight,
                        // B alternating between two pairs so that no instruction overwrites an operand of the one before it).
                        // Wait states by hand -- the compiler's hazard recogniser does not look inside: a dependent
                        // v_mfma_f64_4x4x4 needs 4 after the one that wrote its accumulator (two v_cndmask + s_nop 1), its
                        // result 9+ before anything else reads it (the s_nop at the end).
                        uint32_t ad = (uint32_t)(size_t)pa;
                        const uint32_t adb = ad - 8u * (4u + 3u * (lane & 3));        // d[-4 + k - 4j]: B of the step group before the chunk
                        uint32_t n4 = kend >> 4, n1 = (kend >> 2) & 3;
                        const u64 m0 = 0x1111111111111111ull;                          // lanes with j = 0
#define FG_DPPQ " quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\t"
#define FG_STEP(a0, a1, s0, s1, d0, d1) \
    "v_cndmask_b32_dpp v" #d0 ", v" #s0 ", v" #a0 ", vcc" FG_DPPQ \
    "v_cndmask_b32_dpp v" #d1 ", v" #s1 ", v" #a1 ", vcc" FG_DPPQ \
    "s_nop 1\n\t" \
    "v_mfma_f64_4x4x4_4b_f64 %[acc], v[" #a0 ":" #a1 "], v[" #d0 ":" #d1 "], %[acc]\n\t"
#define FG_RD4(r0, r1, r2, r3, r4, r5, r6, r7, o) \
    "ds_read_b64 v[" #r0 ":" #r1 "], %[ad] offset:" #o "\n\tds_read_b64 v[" #r2 ":" #r3 "], %[ad] offset:" #o "+32\n\t" \
    "ds_read_b64 v[" #r4 ":" #r5 "], %[ad] offset:" #o "+64\n\tds_read_b64 v[" #r6 ":" #r7 "], %[ad] offset:" #o "+96\n\t"
#define FG_RD2(r0, r1, r2, r3, o) \
    "ds_read_b64 v[" #r0 ":" #r1 "], %[ad] offset:" #o "\n\tds_read_b64 v[" #r2 ":" #r3 "], %[ad] offset:" #o "+32\n\t"
#define FG_EVEN(a0, a1) FG_STEP(a0, a1, 56, 57, 58, 59)
#define FG_ODD(a0, a1) FG_STEP(a0, a1, 58, 59, 56, 57)
#if FGX_AUTOC_RD2
                        // (round 5: A of two groups of TWO instructions in flight -- v40..v47 -- instead of two groups of four: eight
                        // registers fewer are out of the compiler's reach, and the kernel no longer spills -- 100 bytes of scratch a lane
                        // were 45 MB written and about as much read back per launch)
                        asm volatile(
                            "s_mov_b64 vcc, %[m0]\n\t"
                            "ds_read_b64 v[56:57], %[adb]\n\t"
                            "s_cmp_eq_u32 %[n4], 0\n\t"
                            "s_cbranch_scc1 2f\n\t"
                            FG_RD2(40, 41, 42, 43, 0)
                            "1:\n\t"
                            FG_RD2(44, 45, 46, 47, 64)
                            "s_waitcnt lgkmcnt(2)\n\t"
                            FG_EVEN(40, 41) FG_ODD(42, 43)
                            "v_add_u32 %[ad], 0x80, %[ad]\n\t"
                            "s_sub_u32 %[n4], %[n4], 1\n\t"
                            "s_cmp_eq_u32 %[n4], 0\n\t"
                            "s_cbranch_scc1 5f\n\t"
                            FG_RD2(40, 41, 42, 43, 0)
                            "s_waitcnt lgkmcnt(2)\n\t"
                            FG_EVEN(44, 45) FG_ODD(46, 47)
                            "s_branch 1b\n\t"
                            "5:\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            FG_EVEN(44, 45) FG_ODD(46, 47)
                            "2:\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cmp_eq_u32 %[n1], 0\n\t"
                            "s_cbranch_scc1 4f\n\t"
                            "3:\n\t"
                            "ds_read_b64 v[40:41], %[ad]\n\t"
                            "v_add_u32 %[ad], 32, %[ad]\n\t"
                            "s_sub_u32 %[n1], %[n1], 1\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            FG_EVEN(40, 41)
                            "v_mov_b32 v56, v58\n\t"
                            "v_mov_b32 v57, v59\n\t"
                            "s_nop 1\n\t"
                            "s_cmp_eq_u32 %[n1], 0\n\t"
                            "s_cbranch_scc0 3b\n\t"
                            "4:\n\t"
                            "s_nop 7\n\t"
                            "s_nop 3"
                            : [acc] "+v"(acc), [ad] "+v"(ad), [n4] "+s"(n4), [n1] "+s"(n1)
                            : [adb] "v"(adb), [m0] "s"(m0)
                            : "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v56", "v57", "v58", "v59");
#else
                        asm volatile(
                            "s_mov_b64 vcc, %[m0]\n\t"
                            "ds_read_b64 v[56:57], %[adb]\n\t"
                            "s_cmp_eq_u32 %[n4], 0\n\t"
                            "s_cbranch_scc1 2f\n\t"
                            FG_RD4(40, 41, 42, 43, 44, 45, 46, 47, 0)
                            "1:\n\t"
                            FG_RD4(48, 49, 50, 51, 52, 53, 54, 55, 128)
                            "s_waitcnt lgkmcnt(4)\n\t"
                            FG_EVEN(40, 41) FG_ODD(42, 43) FG_EVEN(44, 45) FG_ODD(46, 47)
                            "v_add_u32 %[ad], 0x80, %[ad]\n\t"
                            "s_sub_u32 %[n4], %[n4], 1\n\t"
                            "s_cmp_eq_u32 %[n4], 0\n\t"
                            "s_cbranch_scc1 2f\n\t"
                            FG_RD4(40, 41, 42, 43, 44, 45, 46, 47, 128)
                            "s_waitcnt lgkmcnt(4)\n\t"
                            FG_EVEN(48, 49) FG_ODD(50, 51) FG_EVEN(52, 53) FG_ODD(54, 55)
                            "v_add_u32 %[ad], 0x80, %[ad]\n\t"
                            "s_sub_u32 %[n4], %[n4], 1\n\t"
                            "s_cmp_eq_u32 %[n4], 0\n\t"
                            "s_cbranch_scc0 1b\n\t"
                            "2:\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "s_cmp_eq_u32 %[n1], 0\n\t"
                            "s_cbranch_scc1 4f\n\t"
                            "3:\n\t"
                            "ds_read_b64 v[40:41], %[ad]\n\t"
                            "v_add_u32 %[ad], 32, %[ad]\n\t"
                            "s_sub_u32 %[n1], %[n1], 1\n\t"
                            "s_waitcnt lgkmcnt(0)\n\t"
                            FG_EVEN(40, 41)
                            "v_mov_b32 v56, v58\n\t"
                            "v_mov_b32 v57, v59\n\t"
                            "s_nop 1\n\t"
                            "s_cmp_eq_u32 %[n1], 0\n\t"
                            "s_cbranch_scc0 3b\n\t"
                            "4:\n\t"
                            "s_nop 7\n\t"
                            "s_nop 3"
                            : [acc] "+v"(acc), [ad] "+v"(ad), [n4] "+s"(n4), [n1] "+s"(n1)
                            : [adb] "v"(adb), [m0] "s"(m0)
                            : "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
                              "v54", "v55", "v56", "v57", "v58", "v59");
#endif
#undef FG_RD2
#undef FG_EVEN
#undef FG_ODD
#undef FG_RD4
#undef FG_STEP
#undef FG_DPPQ
                    }
                    wave_lds_fence();
                    if (k0 + kn < vec_len && !FGX_DBG(103)) {
                        double t[(NC * FGP_DH + 63) / 64];
#pragma unroll
                        for (int u = 0; u < (NC * FGP_DH + 63) / 64; u++) {
                            const uint32_t j = u * 64 + lane;
                            t[u] = (j < NC * FGP_DH) ? dbuf[(j / FGP_DH) * FGP_CSTR + FGP_CK + (j % FGP_DH)] : 0.0;
                        }
                        wave_lds_fence();
#pragma unroll
                        for (int u = 0; u < (NC * FGP_DH + 63) / 64; u++) {
                            const uint32_t j = u * 64 + lane;
                            if (j < NC * FGP_DH) dbuf[(j / FGP_DH) * FGP_CSTR + (j % FGP_DH)] = t[u];
                        }
                        wave_lds_fence();
                    }
                }
                {
                    const uint32_t oc = (lane >> 2) & 3;
                    const int ol = 4 * (int)(lane & 3) - (int)(lane >> 4);
                    if (oc < (uint32_t)NC && ol >= 0 && ol <= (int)mo) autoc[(oc * P.nvec + nv) * (MAXO + 1) + ol] = acc;
                }
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
            if (P.apod_parts < 2) more = false;
            else if (nv == 1 && vb_ == 1) { vb_ = 2; vc_ = 0; }
            else {
                if (vb_ == 2) { if (vc_ == 0) vc_ = 2; else { vc_ = 0; vb_++; } }
                else if (vc_ < 2 * vb_ - 1) vc_++;
                else { vc_ = 0; vb_++; }
                if (vb_ > P.apod_parts) more = false;
            }
        }
    }
    // ---- wasted bits per candidate (libFLAC shifts them out before any analysis; the autocorrelation of the shifted signal
    // is the one computed here times 2^-2w, exactly)
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t o = wave_or32(orv[c]);
        uint32_t w = o ? (uint32_t)__builtin_ctz(o) : 0;
        const uint32_t nominal = P.bps + ((MS && c == 3) ? 1u : 0u);
        if (w > nominal) w = nominal;
        // (stream_encoder.c get_wasted_bits_wide_: an all-zero 33-bit side channel counts ONE wasted bit -- its constant then fits
        // the 32-bit field)
        if (MS && c == 3 && P.bps == 32 && o == 0) w = 1;
        // (bit 8: every sample of the candidate is zero -- libFLAC counts no wasted bits then, and the 32-bit path must not take
        // that 0 for 'no bits to spare')
        if (lane == 0) { wl[c] = w; B.wasted[bi * NC + c] = w | (o ? 0u : 0x100u); }
    }
    if (lane == 0) B.nv[bi] = nv;
    wave_lds_fence();
    if (mo > 0) {
        const uint32_t per = P.nvec * (MAXO + 1);
        for (uint32_t j = lane; j < (uint32_t)NC * per; j += 64) {
            const uint32_t c = j / per;
            double a = autoc[j];
            const uint32_t w = wl[c];
            if (w) a = ldexp(a, -2 * (int)w);
            B.autoc[(size_t)bi * NC * per + j] = a;
            if (mydbg) {
                const uint32_t r = j % per, v = r / (MAXO + 1), ll = r % (MAXO + 1);
                if (v < nv && ll <= mo) mydbg->cand[c].autoc[v][ll] = a;
            }
        }
        if (mydbg && lane < NC) mydbg->cand[lane].nvec = nv;
    }
}

// ================================================================================================ K2': autocorrelation of a FEW blocks
// (round 6) The call pyFLAC exists for is StreamEncoder.process() with one block: there fg_pipe_autoc_kernel is ONE wave that fetches,
// converts, windows and stores a chunk and then runs its chain of dependent matrix instructions, with the next chunk's samples
// requested one chunk (half a microsecond) ahead -- 46 of the call's 70 us of serial GPU work (round 5).  Here a workgroup works on one
// block.  Wave 0 issues nothing but the chains (the same instructions in the same order: the sums are libFLAC's).  Waves 1 ..
// FGP_A1S stage: wave s takes the chunks s - 1, s - 1 + FGP_A1S, ... -- requests the samples of its next chunk as soon as it has stored
// one, i.e. FGP_A1S chunks ahead, with nothing else of its own in flight: the compiler's wait in front of the conversion is for exactly
// those loads (a queue of chunks in the registers of one wave makes it wait for all or copy registers a load has not reached yet).
// The chunks go into three slots in LDS, one behind the other, so that the chain's reads of the fifteen values in front of its chunk
// fall into the slot before (in front of slot 0 a copy of the end of slot 2); the workgroup meets at a barrier a chunk: in phase p
// chunk p is stored while the chain runs over chunk p - 1, whose history in slot (p - 2) % 3 nobody writes.
// Used for launches of up to FGP_A1_MAX blocks (pipe_shape.inc): beyond that the chip is full of one-wave blocks anyway.
// Windows side by side (wpar, levels 6 - 8: subdivide_tukey): the whole-block window, the half-block and the third-block windows are
// independent chains, so a launch of very few blocks gives each its own workgroup (blockIdx.y = which of them; up to six), which
// leaves its vector as it comes out of the chain; fg_pipe_autoc_fix_kernel then makes the punched windows (differences of two
// vectors) and scales everything by the wasted bits, which only the workgroup of the whole-block window knows.
#define FGP_A1S 8
#define FGP_A1_MAX 256
#define FGP_A1RSTR (FGP_DH + 3 * FGP_CK + FGP_SLK)
FGI void pipe_a1_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <bool MS, int NCH, int MAXO>
__global__ void __launch_bounds__(64 * (1 + FGP_A1S))
fg_pipe_autoc1_kernel(const void *pcm, const FgBlockDesc *descs, const float *windows, FgEncParams P, FgPipeBufs B, FgDebugRec *dbg,
                      uint32_t nblocks, uint32_t bi0, uint32_t wpar)
{
    constexpr int NC = MS ? 4 : NCH;
    constexpr uint32_t RSTR = FGP_A1RSTR;                       // a candidate's row: history of slot 0, three slots, zeros
    constexpr uint32_t NT = 64 * (1 + FGP_A1S);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t wv = rfl(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint32_t bi = (bi0 & 0x3FFFFFFFu) + blockIdx.x;
    // (bits 31 / 30 of bi0: as in fg_pipe_autoc_kernel)
    if ((bi0 >> 30) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (B.stamp) B.stamp[0] = wall_clock64();
        if ((bi0 >> 31) && B.guard) { B.guard[0] = 0ull; B.guard[1] = 0x7FF0000000000000ull; B.guard[2] = 0ull; }
    }
    if (bi >= nblocks) return;
    const FgBlockDesc d = descs[bi];
    const uint32_t n = d.n;
    LDS double *ring = (LDS double *)smem;                      // [NC][FGP_DH | 3 x FGP_CK | FGP_SLK]
    LDS double *autoc = ring + NC * RSTR;                       // [NC][nvec][MAXO + 1]
    LDS uint32_t *wl = (LDS uint32_t *)(autoc + NC * P.nvec * (MAXO + 1));   // wasted bits per candidate, then the OR of its samples
    LDS uint32_t *orx = wl + 4;
    // (pointers the compiler can still see to be global: its loads then count in vmcnt alone; through uniform_ptr they are generic, a
    // flat load may be an LDS access, and every wait is for everything)
    const float *const window = windows + d.win_off;
    const unsigned char *const pcmb = (const unsigned char *)pcm + (size_t)d.pcm_off * (NCH == 2 ? (P.pcm_i16 ? 4u : 8u) : (P.pcm_i16 ? 2u : 4u));
    FgDebugRec *mydbg = dbg ? dbg + d.out_slot : nullptr;
    if (threadIdx.x < 4) orx[threadIdx.x] = 0;
    pipe_a1_barrier();                  // (the staging waves OR into these words: at the end, which with no LPC to do is at once)
    uint32_t orv[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) orv[c] = 0;
    // sample g of the block as it lies in memory (a 16-bit pair in one word: taken apart at the point of use), and L, R of it
    auto ldraw = [&](uint32_t g, int32_t &a, int32_t &b) __attribute__((always_inline)) {
        b = 0;
        if (NCH == 2) {
            if (P.pcm_i16) a = ((const int32_t *)pcmb)[g];
            else { const int2 v = ((const int2 *)pcmb)[g]; a = v.x; b = v.y; }
        }
        else {
            // (one channel of an interleaved stream of more channels: the block is a view with a stride, see FgBlockDesc)
            const uint32_t at = g * (d.reserved ? d.reserved : 1u);
            a = P.pcm_i16 ? (int32_t)((const int16_t *)pcmb)[at] : ((const int32_t *)pcmb)[at];
        }
    };
    auto unraw = [&](int32_t &L, int32_t &R) __attribute__((always_inline)) {
        if (NCH == 2 && P.pcm_i16) { R = L >> 16; L = (int32_t)(int16_t)L; }
    };
    // (candidate c of a sample as the float libFLAC windows, and the bits its wasted-bits count looks at: fg_pipe_autoc_kernel)
    auto cval = [&](int c, int32_t L, int32_t R, float &fx, uint32_t &ob) __attribute__((always_inline)) {
        if (MS && P.bps == 32 && c == 2) { const int32_t m_ = (int32_t)(((i64)L + (i64)R) >> 1); fx = (float)m_; ob = (uint32_t)m_; }
        else if (MS && P.bps == 32 && c == 3) { fx = (float)((double)L - (double)R); ob = (uint32_t)L - (uint32_t)R; }
        else {
            const int32_t x = c == 0 ? pcv<MS, 0>(L, R) : c == 1 ? pcv<MS, 1>(L, R) : c == 2 ? pcv<MS, 2>(L, R) : pcv<MS, 3>(L, R);
            fx = (float)x; ob = (uint32_t)x;
        }
    };
    uint32_t nv = 0;
    const uint32_t mo = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
    // (wpar: this workgroup's window among those that run a chain, and where its vector went)
    const uint32_t wsel = wpar ? blockIdx.y : 0xFFFFFFFFu;
    uint32_t realw = 0, myv = 0xFFFFFFFFu;
    if (mo == 0) {
        // no LPC at this level: only the OR of the samples is needed
        if (wv != 0 && (!wpar || wsel == 0)) {
            for (uint32_t i = (wv - 1) * 64 + (uint32_t)lane; i < n; i += 64 * FGP_A1S) {
                int32_t L, R;
                ldraw(i, L, R);
                unraw(L, R);
#pragma unroll
                for (int c = 0; c < NC; c++) { float fx; uint32_t ob; cval(c, L, R, fx, ob); orv[c] |= ob; }
            }
        }
    }
    else {
        const uint32_t mrow = ((lane >> 2) & 3) < (uint32_t)NC ? ((lane >> 2) & 3) : 0;
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
            if (!skip && !punch && wsel != 0xFFFFFFFFu && wsel != realw++) { /* another workgroup's window */ }
            else if (!skip && !punch) {
                myv = nv;
                // the history of the first chunk: zeros
                if (threadIdx.x < (uint32_t)NC * FGP_DH) ring[(threadIdx.x / FGP_DH) * RSTR + (threadIdx.x % FGP_DH)] = 0.0;
                pipe_a1_barrier();
                const uint32_t nch = (vec_len + FGP_CK - 1) / FGP_CK;
                // phase p = 0 .. nch: chunk p is stored (p < nch), the chain runs over chunk p - 1 (p >= 1); a barrier ends it
                if (wv == 0) {
                    double acc = 0.0;
                    for (uint32_t p = 0; p <= nch; p++) {
                        if (p >= 1) {
                            const uint32_t k0 = (p - 1) * FGP_CK;
                            const uint32_t kn = (vec_len - k0) < FGP_CK ? (vec_len - k0) : FGP_CK;
                            const uint32_t kend = (k0 + kn < vec_len) ? FGP_CK : ((kn + 3) & ~3u) + 4;
                            const LDS double *pa = ring + mrow * RSTR + FGP_DH + ((p - 1) % 3u) * FGP_CK + (lane >> 4) - (lane & 3);
                            // (the chain of fg_pipe_autoc_kernel, which explains it)
                            uint32_t ad = (uint32_t)(size_t)pa;
                            const uint32_t adb = ad - 8u * (4u + 3u * (lane & 3));
                            uint32_t n4 = kend >> 4, n1 = (kend >> 2) & 3;
                            const u64 m0 = 0x1111111111111111ull;
#define FG_DPPQ " quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\t"
#define FG_STEP(a0, a1, s0, s1, d0, d1) \
    "v_cndmask_b32_dpp v" #d0 ", v" #s0 ", v" #a0 ", vcc" FG_DPPQ \
    "v_cndmask_b32_dpp v" #d1 ", v" #s1 ", v" #a1 ", vcc" FG_DPPQ \
    "s_nop 1\n\t" \
    "v_mfma_f64_4x4x4_4b_f64 %[acc], v[" #a0 ":" #a1 "], v[" #d0 ":" #d1 "], %[acc]\n\t"
#define FG_RD2(r0, r1, r2, r3, o) \
    "ds_read_b64 v[" #r0 ":" #r1 "], %[ad] offset:" #o "\n\tds_read_b64 v[" #r2 ":" #r3 "], %[ad] offset:" #o "+32\n\t"
#define FG_EVEN(a0, a1) FG_STEP(a0, a1, 56, 57, 58, 59)
#define FG_ODD(a0, a1) FG_STEP(a0, a1, 58, 59, 56, 57)
                            asm volatile(
                                "s_mov_b64 vcc, %[m0]\n\t"
                                "ds_read_b64 v[56:57], %[adb]\n\t"
                                "s_cmp_eq_u32 %[n4], 0\n\t"
                                "s_cbranch_scc1 2f\n\t"
                                FG_RD2(40, 41, 42, 43, 0)
                                "1:\n\t"
                                FG_RD2(44, 45, 46, 47, 64)
                                "s_waitcnt lgkmcnt(2)\n\t"
                                FG_EVEN(40, 41) FG_ODD(42, 43)
                                "v_add_u32 %[ad], 0x80, %[ad]\n\t"
                                "s_sub_u32 %[n4], %[n4], 1\n\t"
                                "s_cmp_eq_u32 %[n4], 0\n\t"
                                "s_cbranch_scc1 5f\n\t"
                                FG_RD2(40, 41, 42, 43, 0)
                                "s_waitcnt lgkmcnt(2)\n\t"
                                FG_EVEN(44, 45) FG_ODD(46, 47)
                                "s_branch 1b\n\t"
                                "5:\n\t"
                                "s_waitcnt lgkmcnt(0)\n\t"
                                FG_EVEN(44, 45) FG_ODD(46, 47)
                                "2:\n\t"
                                "s_waitcnt lgkmcnt(0)\n\t"
                                "s_cmp_eq_u32 %[n1], 0\n\t"
                                "s_cbranch_scc1 4f\n\t"
                                "3:\n\t"
                                "ds_read_b64 v[40:41], %[ad]\n\t"
                                "v_add_u32 %[ad], 32, %[ad]\n\t"
                                "s_sub_u32 %[n1], %[n1], 1\n\t"
                                "s_waitcnt lgkmcnt(0)\n\t"
                                FG_EVEN(40, 41)
                                "v_mov_b32 v56, v58\n\t"
                                "v_mov_b32 v57, v59\n\t"
                                "s_nop 1\n\t"
                                "s_cmp_eq_u32 %[n1], 0\n\t"
                                "s_cbranch_scc0 3b\n\t"
                                "4:\n\t"
                                "s_nop 7\n\t"
                                "s_nop 3"
                                : [acc] "+v"(acc), [ad] "+v"(ad), [n4] "+s"(n4), [n1] "+s"(n1)
                                : [adb] "v"(adb), [m0] "s"(m0)
                                : "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v56", "v57", "v58", "v59");
#undef FG_RD2
#undef FG_EVEN
#undef FG_ODD
#undef FG_STEP
#undef FG_DPPQ
                        }
                        pipe_a1_barrier();
                    }
                    {
                        const uint32_t oc = (lane >> 2) & 3;
                        const int ol = 4 * (int)(lane & 3) - (int)(lane >> 4);
                        if (oc < (uint32_t)NC && ol >= 0 && ol <= (int)mo) autoc[(oc * P.nvec + nv) * (MAXO + 1) + ol] = acc;
                    }
                    wave_lds_fence();
                }
                else {
                    // samples lane and lane + 64 of the chunk that starts at k0, as they lie in memory, and their window values.  No
                    // branch around the loads -- behind one the compiler waits for them on the spot --: past the vector's end the last
                    // valid sample is fetched again, and the conversion below puts a zero in its place.
                    const uint32_t lim = part ? 2 * part : vec_len;              // (>= 1: a punched or skipped window never comes here)
                    const uint32_t wjump = part ? n - 2 * part : 0u;             // second half of a partial window: the window's falling end
                    // (one copy of the loop per sample format: where the two forms of a load meet, the compiler waits for the other's)
                    auto staging = [&](auto I16) __attribute__((always_inline)) {
                    constexpr bool i16 = decltype(I16)::value;
                    float w0, w1;
                    int32_t l0, r0, l1, r1;
                    auto ld = [&](uint32_t g, int32_t &a, int32_t &b) __attribute__((always_inline)) {
                        b = 0;
                        if (NCH == 2) {
                            if (i16) a = ((const int32_t *)pcmb)[g];
                            else { const int2 v = ((const int2 *)pcmb)[g]; a = v.x; b = v.y; }
                        }
                        else {
                            const uint32_t at = g * (d.reserved ? d.reserved : 1u);
                            a = i16 ? (int32_t)((const int16_t *)pcmb)[at] : ((const int32_t *)pcmb)[at];
                        }
                    };
                    auto fetch = [&](uint32_t k0) __attribute__((always_inline)) {
                        const uint32_t a0 = k0 + (uint32_t)lane, a1 = a0 + 64;
                        const uint32_t i0 = a0 < lim ? a0 : lim - 1, i1 = a1 < lim ? a1 : lim - 1;
                        w0 = window[i0 + ((part != 0 && i0 >= part) ? wjump : 0u)];
                        w1 = window[i1 + ((part != 0 && i1 >= part) ? wjump : 0u)];
                        ld(sh + i0, l0, r0);
                        ld(sh + i1, l1, r1);
                    };
                    uint32_t mine = wv - 1;                                      // the next chunk this wave stores
                    fetch(mine * FGP_CK);
                    for (uint32_t p = 0; p <= nch; p++) {
                        if (p == mine && p < nch) {
                            const uint32_t base = FGP_DH + (p % 3u) * FGP_CK;
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                int32_t L = h ? l1 : l0, R = h ? r1 : r0;
                                const float w = h ? w1 : w0;
                                if (NCH == 2 && i16) { R = L >> 16; L = (int32_t)(int16_t)L; }
                                const uint32_t j = (uint32_t)lane + 64u * h;
                                const bool zero = (p * FGP_CK + j) >= lim;
#pragma unroll
                                for (int c = 0; c < NC; c++) {
                                    float fx;
                                    uint32_t ob;
                                    cval(c, L, R, fx, ob);
                                    if (nv == 0) orv[c] |= zero ? 0u : ob;
                                    const float dd = zero ? 0.0f : fx * w;
                                    const double v = (double)dd;
                                    ring[c * RSTR + base + j] = v;
                                    // (the end of slot 2 once more in front of slot 0, where the chain on the next chunk looks for its history)
                                    if (h == 1 && p % 3u == 2 && j >= FGP_CK - FGP_DH) ring[c * RSTR + (j - (FGP_CK - FGP_DH))] = v;
                                }
                            }
                            // (the vector's last chunk: the chain runs up to seven steps past it -- zeros; in the slot behind, or in the
                            // slack behind slot 2: nobody stores a chunk there any more)
                            if (p + 1 == nch && lane < FGP_SLK) {
#pragma unroll
                                for (int c = 0; c < NC; c++) ring[c * RSTR + base + FGP_CK + lane] = 0.0;
                            }
                            mine += FGP_A1S;
                            fetch(mine * FGP_CK);
                        }
                        pipe_a1_barrier();
                    }
                    };
                    if (P.pcm_i16) staging(std::integral_constant<bool, true>()); else staging(std::integral_constant<bool, false>());
                }
            }
            else if (punch) {
                // root - previous partial for lags < mo; lag mo keeps the partial (upstream quirk)
                if (wv == 0 && !wpar) {
                    const uint32_t total = (uint32_t)NC * (mo + 1);
                    for (uint32_t j = lane; j < total; j += 64) {
                        const uint32_t c = j / (mo + 1), ll = j % (mo + 1);
                        LDS double *base = autoc + c * P.nvec * (MAXO + 1);
                        const double prev = base[(nv - 1) * (MAXO + 1) + ll];
                        base[nv * (MAXO + 1) + ll] = (ll < mo) ? base[ll] - prev : prev;
                    }
                    wave_lds_fence();
                }
            }
            if (!skip) nv++;
            if (P.apod_parts < 2) more = false;
            else if (nv == 1 && vb_ == 1) { vb_ = 2; vc_ = 0; }
            else {
                if (vb_ == 2) { if (vc_ == 0) vc_ = 2; else { vc_ = 0; vb_++; } }
                else if (vc_ < 2 * vb_ - 1) vc_++;
                else { vc_ = 0; vb_++; }
                if (vb_ > P.apod_parts) more = false;
            }
        }
    }
    // ---- the OR of every candidate's samples: from the staging waves to the chain wave, which finishes the block alone
    if (wv != 0) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint32_t o = wave_or32(orv[c]);
            if (lane == 0 && o) __hip_atomic_fetch_or(&orx[c], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    pipe_a1_barrier();
    if (wv != 0) return;
    if (wpar) {
        // this window's vector as the chain left it (fg_pipe_autoc_fix_kernel does the rest); the whole-block window's workgroup has
        // seen every sample: the wasted bits and the number of vectors are its to say
        if (myv != 0xFFFFFFFFu && mo > 0) {
            for (uint32_t j = lane; j < (uint32_t)NC * (MAXO + 1); j += 64) {
                const uint32_t c = j / (MAXO + 1), ll = j % (MAXO + 1);
                B.autoc[((size_t)bi * NC + c) * P.nvec * (MAXO + 1) + (size_t)myv * (MAXO + 1) + ll] = autoc[(c * P.nvec + myv) * (MAXO + 1) + ll];
            }
        }
        if (wsel != 0) return;
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t o = orx[c];
        uint32_t w = o ? (uint32_t)__builtin_ctz(o) : 0;
        const uint32_t nominal = P.bps + ((MS && c == 3) ? 1u : 0u);
        if (w > nominal) w = nominal;
        if (MS && c == 3 && P.bps == 32 && o == 0) w = 1;                          // (get_wasted_bits_wide_, see fg_pipe_autoc_kernel)
        if (lane == 0) { wl[c] = w; B.wasted[bi * NC + c] = w | (o ? 0u : 0x100u); }
    }
    if (lane == 0) B.nv[bi] = nv;
    wave_lds_fence();
    if (mo > 0 && !wpar) {
        const uint32_t per = P.nvec * (MAXO + 1);
        for (uint32_t j = lane; j < (uint32_t)NC * per; j += 64) {
            const uint32_t c = j / per;
            double a = autoc[j];
            const uint32_t w = wl[c];
            if (w) a = ldexp(a, -2 * (int)w);
            B.autoc[(size_t)bi * NC * per + j] = a;
            if (mydbg) {
                const uint32_t r = j % per, v = r / (MAXO + 1), ll = r % (MAXO + 1);
                if (v < nv && ll <= mo) mydbg->cand[c].autoc[v][ll] = a;
            }
        }
        if (mydbg && lane < NC) mydbg->cand[lane].nvec = nv;
    }
}

// (behind fg_pipe_autoc1_kernel with its windows side by side: the punched windows and the scaling by the wasted bits, as the tail of
// that kernel does them when one workgroup has all the vectors.  One wave a block.)
template <int MAXO>
__global__ void __launch_bounds__(64)
fg_pipe_autoc_fix_kernel(const FgBlockDesc *descs, FgEncParams P, FgPipeBufs B, uint32_t NC, uint32_t nblocks, uint32_t bi0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LDS double *a = (LDS double *)smem;                 // [NC][nvec][MAXO + 1]
    const uint32_t bi = bi0 + blockIdx.x;
    if (bi >= nblocks) return;
    const uint32_t n = descs[bi].n;
    const uint32_t mo = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
    if (mo == 0) return;
    const int lane = threadIdx.x;
    const uint32_t per = P.nvec * (MAXO + 1), total = NC * per;
    double *g = B.autoc + (size_t)bi * total;
    for (uint32_t j = lane; j < total; j += 64) a[j] = g[j];
    wave_lds_fence();
    // the windows in the order the autocorrelation kernels count them (fg_pipe_autoc_kernel); the punched ones in place
    uint32_t nv = 0, vb_ = 1, vc_ = 0;
    bool more = true;
    while (more) {
        bool punch = false, skip = false;
        if (nv > 0) {
            if (n / vb_ <= 32) skip = true;
            else if (vc_ & 1) punch = true;
        }
        if (punch) {
            for (uint32_t j = lane; j < NC * (mo + 1); j += 64) {
                const uint32_t c = j / (mo + 1), ll = j % (mo + 1);
                LDS double *base = a + c * per;
                const double prev = base[(nv - 1) * (MAXO + 1) + ll];
                base[nv * (MAXO + 1) + ll] = (ll < mo) ? base[ll] - prev : prev;
            }
            wave_lds_fence();
        }
        if (!skip) nv++;
        if (P.apod_parts < 2) more = false;
        else if (nv == 1 && vb_ == 1) { vb_ = 2; vc_ = 0; }
        else {
            if (vb_ == 2) { if (vc_ == 0) vc_ = 2; else { vc_ = 0; vb_++; } }
            else if (vc_ < 2 * vb_ - 1) vc_++;
            else { vc_ = 0; vb_++; }
            if (vb_ > P.apod_parts) more = false;
        }
    }
    for (uint32_t c = 0; c < NC; c++) {
        // (2^-2w: a power of two, exact)
        const uint32_t w = rfl(B.wasted[bi * NC + c]) & 0xFFu;
        const double sc = __hiloint2double((int)((1023u - 2u * w) << 20), 0);
        for (uint32_t j = lane; j < per; j += 64) g[c * per + j] = w ? a[c * per + j] * sc : a[c * per + j];
    }
}

// ---- correctly rounded log for the near-ties of the order guess (double-double arithmetic; rare path)
// libFLAC picks the LPC order by comparing bits = 0.5 * log(scale * err) / ln 2 * (n - o) + o * overhead across orders with a
// strict <.  glibc's log and the device's differ in the last bit now and then, which can only matter when two orders come
// out within ~1e-10 bits of each other; such lanes repeat the comparison with a log that is exact to 2^-100 and rounded
// once (what glibc returns in all but its documented 0.02-ulp band around rounding boundaries).
struct pdd { double hi, lo; };
FGI pdd pdd_sum(double a, double b) { const double s = a + b, bb = s - a; pdd r; r.hi = s; r.lo = (a - (s - bb)) + (b - bb); return r; }
FGI pdd pdd_add(pdd x, pdd y)
{
    pdd s = pdd_sum(x.hi, y.hi);
    s.lo += x.lo + y.lo;
    return pdd_sum(s.hi, s.lo);
}
FGI pdd pdd_mul(pdd x, pdd y)
{
    const double p = x.hi * y.hi;
    double e = fma(x.hi, y.hi, -p);
    e += x.hi * y.lo + x.lo * y.hi;
    return pdd_sum(p, e);
}
FGI pdd pdd_div(pdd x, pdd y)
{
    const double q1 = x.hi / y.hi;
    // r = x - q1 * y
    const double p = q1 * y.hi;
    const double e = fma(q1, y.hi, -p);
    const double r = ((x.hi - p) - e) + x.lo - q1 * y.lo;
    const double q2 = r / y.hi;
    return pdd_sum(q1, q2);
}
FGI double plog_cr(double x)       // x > 0, finite
{
    int k;
    double m = frexp(x, &k);                         // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; k--; }      // m in [1/sqrt 2, sqrt 2)
    pdd one; one.hi = 1.0; one.lo = 0.0;
    const pdd num = pdd_sum(m, -1.0), den = pdd_sum(m, 1.0);
    const pdd sv = pdd_div(num, den);
    const pdd s2 = pdd_mul(sv, sv);
    // sum_{j=0}^{21} s2^j / (2j + 1), Horner from the top
    pdd t; t.hi = 1.0 / 43.0; t.lo = fma(-t.hi, 43.0, 1.0) / 43.0;
    for (int j = 20; j >= 0; j--) {
        pdd c; const double dd = (double)(2 * j + 1);
        c.hi = 1.0 / dd; c.lo = fma(-c.hi, dd, 1.0) / dd;
        t = pdd_add(pdd_mul(t, s2), c);
    }
    pdd lm = pdd_mul(sv, t);
    lm.hi *= 2.0; lm.lo *= 2.0;
    pdd ln2; ln2.hi = 0.6931471805599453094; ln2.lo = 2.3190468138462996e-17;
    pdd kk; kk.hi = (double)k; kk.lo = 0.0;
    const pdd r = pdd_add(pdd_mul(kk, ln2), lm);
    return r.hi + r.lo;
}
FGI double p_ebps_cr(double e, double scale)
{
    if (e > 0.0) {
        const double bb = 0.5 * plog_cr(scale * e) / FG_LN2;
        return bb >= 0.0 ? bb : 0.0;
    }
    else if (e < 0.0) return 1e32;
    return 0.0;
}

// ================================================================================================ K3: Levinson-Durbin, order guess, quantiser
// lane = (block, candidate, vector).  lres = order | prec<<8 | (shift&255)<<16 | ok<<24 | ran<<25.  Same operations in the
// same order as lpc.c (the file is compiled with -ffp-contract=off), so the same doubles.
template <int MAXO>
__global__ void __launch_bounds__(64)
fg_pipe_levinson_kernel(const FgBlockDesc *descs, FgEncParams P, FgPipeBufs B, uint32_t nblocks, uint32_t NC, uint32_t ms, double guard_thr,
                        uint32_t bi0)
{
    const uint32_t per = NC * P.nvec;
    const uint32_t idx = bi0 * per + blockIdx.x * 64 + threadIdx.x;     // (blocks [bi0, nblocks))
    if (idx >= nblocks * per) return;
    const uint32_t bi = idx / per, r_ = idx % per, c = r_ / P.nvec, v = r_ % P.nvec;
    const uint32_t n = descs[bi].n;
    const uint32_t nv = B.nv[bi];
    const uint32_t mo = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
    const double *Ap = B.autoc + (size_t)idx * (MAXO + 1);
    bool on = v < nv && mo > 0;
    double A[MAXO + 1];
#pragma unroll
    for (int j = 0; j <= MAXO; j++) A[j] = (on && (uint32_t)j <= mo) ? Ap[j] : 0.0;
    if (on && A[0] == 0.0) on = false;
    if (!on) {
#pragma unroll
        for (int j = 0; j <= MAXO; j++) A[j] = 0.0;
    }
    const uint32_t sb = P.bps + ((ms && c == 3) ? 1u : 0u) - (B.wasted[bi * NC + c] & 0xFFu);
    const double a0 = on ? A[0] : 1.0;
    const uint32_t overhead = sb + P.qlp_precision;
    const double scale = 0.5 / (double)n;
    double er = a0, bestb = 4294967295.0;
    double lp[MAXO], keep[MAXO], err2 = a0;
    double erv[MAXO], bitsv[MAXO];              // error after each order and its bit estimate (for the near-tie guard)
#pragma unroll
    for (int j = 0; j < MAXO; j++) { lp[j] = 0.0; keep[j] = 0.0; erv[j] = 0.0; bitsv[j] = -1.0; }
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
                const double bits = p_ebps(er, scale) * (double)(n - o) + (double)(o * overhead);
                erv[i] = er; bitsv[i] = bits;
                if (bits < bestb) { besti = i; bestb = bits; better = true; }
                if (er == 0.0) stopped = true;
            }
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
    // ---- near-tie guard of the order guess (see plog_cr): margin = distance of the runner-up from the winner
    if (on) {
        double margin = 1e300;
#pragma unroll
        for (int i = 0; i < MAXO; i++)
            if ((uint32_t)i != besti && bitsv[i] >= 0.0) { const double dm = bitsv[i] - bestb; margin = dm < margin ? dm : margin; }
        if (margin < 1e300 && B.guard) atomicMin(&B.guard[1], (unsigned long long)__double_as_longlong(margin < 0.0 ? 0.0 : margin));
        if (margin < guard_thr) {
            if (B.guard) atomicAdd(&B.guard[0], 1ull);
            double bb2 = 4294967295.0;
            uint32_t b2 = 0;
#pragma unroll
            for (int i = 0; i < MAXO; i++) {
                if (bitsv[i] >= 0.0) {
                    const uint32_t o = i + 1;
                    const double bits = p_ebps_cr(erv[i], scale) * (double)(n - o) + (double)(o * overhead);
                    if (bits < bb2) { b2 = i; bb2 = bits; }
                }
            }
            if (b2 != besti) {
                // the exact comparison picks another order: its coefficient set comes from running the recursion again
                besti = b2;
                double er_ = a0;
#pragma unroll
                for (int j = 0; j < MAXO; j++) lp[j] = 0.0;
#pragma unroll
                for (int i = 0; i < MAXO; i++) {
                    if ((uint32_t)i <= besti && (uint32_t)i < mo) {
                        double r = -A[i + 1];
#pragma unroll
                        for (int j = 0; j < i; j++) r -= lp[j] * A[i - j];
                        r /= er_;
                        lp[i] = r;
#pragma unroll
                        for (int j = 0; j < (i >> 1); j++) {
                            const double tmp = lp[j], t2 = lp[i - 1 - j];
                            lp[j] = tmp + r * t2;
                            lp[i - 1 - j] = t2 + r * tmp;
                        }
                        if (i & 1) { const double t = lp[i >> 1]; lp[i >> 1] = t + t * r; }
                        er_ *= (1.0 - r * r);
                    }
                }
#pragma unroll
                for (int j = 0; j < MAXO; j++) keep[j] = (uint32_t)j <= besti ? lp[j] : 0.0;
                err2 = er_;
            }
        }
    }
    const uint32_t ostar = besti + 1;
    float lpf[MAXO];
#pragma unroll
    for (int j = 0; j < MAXO; j++) lpf[j] = (float)(-keep[j]);
    uint32_t result = 0;
    int32_t qv_[MAXO];
#pragma unroll
    for (int j = 0; j < MAXO; j++) qv_[j] = 0;
    if (on) {
        bool ok = !(p_ebps(err2, 0.5 / (double)(n - ostar)) >= (double)sb);
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
                        qv_[j] = qv;
                    }
                }
                if (neg) shift = 0;
            }
        }
        result = ostar | (prec << 8) | (((uint32_t)shift & 0xFF) << 16) | ((ok ? 1u : 0u) << 24) | (1u << 25);
    }
#pragma unroll
    for (int j = 0; j < MAXO; j++) B.qres[(size_t)idx * MAXO + j] = qv_[j];
    B.lres[idx] = result;
}

// ================================================================================================ lane geometry
// Regular blocks (n a multiple of the lane count, 16 or more samples a lane): lane l owns samples [l seg, (l + 1) seg), row l of
// the LDS staging holds them, rows are seg + PADE elements apart.  Everything else -- the tail block of nearly every real
// stream, odd block sizes, short blocks -- takes the RAGGED geometry: the lanes are grouped by the partitions of the finest
// partition order the block allows (pm = min(max order, trailing zeros of n): G = 2^pm groups of lpg = lanes / G lanes and
// S = n / G samples), and the S samples of a group are spread over its first A = min(lpg, S / 16) lanes as evenly as they go:
// `base` samples each, one more for the first `extra` lanes.  So every working lane has 16 or more samples (the predictor
// history of a lane lies in ONE other lane, and the warm-up samples lie in lane 0), a partition of any order is a run of whole
// lanes exactly as in the regular geometry -- the Rice search and the packer's partition logic do not change --, and lanes
// A .. lpg - 1 of a group idle.  Row l of the staging holds lane l's samples, rows are base + 1 + PADE apart.
// elements of one channel's staging area: the regular rows need sig_stride + 2 a lane, the ragged ones 64 rows of up to 31 + 3
FGI uint32_t pipe_rows_elems(uint32_t sig_stride, uint32_t lanes) { return (sig_stride > 2048u ? sig_stride : 2048u) + 2u * lanes + 128u; }

struct PipeGeo {
    uint32_t rag;               // 0: regular
    uint32_t lpgs;              // log2(lanes per group)
    uint32_t S, A, base, extra, rstr;
};
FGI PipeGeo pipe_geo(uint32_t n, uint32_t lanes_log2, uint32_t max_po, uint32_t pade)
{
    PipeGeo g;
    const uint32_t lanes = 1u << lanes_log2;
    g.rag = ((n & (lanes - 1)) != 0 || (n >> lanes_log2) < 16) ? 1u : 0u;
    uint32_t pm = 0;
    { uint32_t b = n; while (!(b & 1) && pm < 15) { pm++; b >>= 1; } }
    if (pm > max_po) pm = max_po;
    if (pm > 6) pm = 6;
    g.lpgs = lanes_log2 - pm;
    g.S = n >> pm;
    const uint32_t lpg = 1u << g.lpgs;
    uint32_t A = g.S / 16;
    if (A < 1) A = 1;
    if (A > lpg) A = lpg;
    g.A = A; g.base = g.S / A; g.extra = g.S - g.base * A;
    g.rstr = g.rag ? g.base + 1 + pade : (n >> lanes_log2) + pade;
    return g;
}
FGI PipeGeo pipe_geo_regular(uint32_t seg, uint32_t pade)
{
    PipeGeo g;
    g.rag = 0; g.lpgs = 0; g.S = 1; g.A = 1; g.base = 1; g.extra = 0; g.rstr = seg + pade;
    return g;
}
// what a lane owns, and where the samples in front of its first one lie
struct PipeLane { uint32_t len, act, hasprev, prow, plen; };
template <bool RAG>
FGI PipeLane pipe_lane(const PipeGeo &g, uint32_t Lg, uint32_t seg)
{
    PipeLane L;
    if (!RAG) { L.len = seg; L.act = 1; L.hasprev = Lg > 0; L.prow = Lg - 1; L.plen = seg; return L; }
    const uint32_t grp = Lg >> g.lpgs, w = Lg & ((1u << g.lpgs) - 1);
    L.act = w < g.A ? 1u : 0u;
    L.len = L.act ? g.base + (w < g.extra ? 1u : 0u) : 0u;
    L.hasprev = Lg > 0;
    if (w > 0) { L.prow = Lg - 1; L.plen = g.base + ((w - 1) < g.extra ? 1u : 0u); }
    else { L.prow = ((grp - 1) << g.lpgs) + g.A - 1; L.plen = g.base + ((g.A - 1) < g.extra ? 1u : 0u); }
    if (!L.act) { L.prow = 0; L.plen = g.base; }       // (idle lanes read row 0: anything that is there)
    return L;
}
// element index of sample i of the block in the staging (ragged geometry)
FGI uint32_t pipe_rag_addr(const PipeGeo &g, uint32_t i, uint32_t mS, uint32_t mB1, uint32_t mB)
{
    const uint32_t grp = __umulhi(i, mS) , r = i - grp * g.S;          // i / S through the magic multiplier (exact for i < 2^16 * S)
    const uint32_t cut = g.extra * (g.base + 1);
    uint32_t w, col;
    if (r < cut) { w = __umulhi(r, mB1); col = r - w * (g.base + 1); }
    else { const uint32_t r2 = r - cut; const uint32_t w2 = __umulhi(r2, mB); w = g.extra + w2; col = r2 - w2 * g.base; }
    return ((grp << g.lpgs) + w) * g.rstr + col;
}

// 32-bit streams.  The forms of the 17..25-bit path (24-bit multiplies for the candidate, fp64 FIR, 32-bit error terms) hold for a
// 32-bit stream whose channels share at least eight wasted bits -- 24-bit material in a 32-bit container, what pyFLAC makes of a
// 24-bit WAV file (soundfile reads it as left-justified int32, pyflac/encoder.py:109 sets 32 bits per sample): the samples are
// staged shifted down by the shared count `pre`, every later shift is by (wasted - pre), and all widths -- subframe bits per
// sample, warm-up fields, the wasted-bits field itself -- come out as libFLAC's, which shifts by the whole count.  A block whose
// channels share fewer bits (true 32-bit content) goes to the generic kernel (ok = false).  Channels that are all zero have no
// say.  B.wasted: bits 0..7 the count, bit 8 all-zero (fg_pipe_autoc_kernel).
template <int NCH, int NC>
FGI uint32_t pipe_preshift(const FgEncParams &P, const FgPipeBufs &B, uint32_t bi, bool &ok)
{
    ok = true;
    if (P.bps != 32) return 0;
    uint32_t m = 64;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint32_t w = rfl(B.wasted[bi * NC + c]);
        if (!(w & 0x100u)) m = (w & 0xFFu) < m ? (w & 0xFFu) : m;
    }
    if (m == 64) m = 8;
    if (m < 8) { ok = false; return 0; }
    // (every candidate must fit the 25-bit forms once its own wasted bits are gone: mid and side may keep more than the channels)
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t w = rfl(B.wasted[bi * NC + c]);
        const uint32_t nominal = 32u + ((NC == 4 && c == 3) ? 1u : 0u);
        if (!(w & 0x100u) && nominal - (w & 0xFFu) > 25u) { ok = false; return 0; }
    }
    return m > 16 ? 16 : m;
}

// True 32-bit content (round 4): candidate values of up to 33 bits as doubles -- L, R as they are, mid = floor((L + R) / 2), side =
// L - R, all exact in fp64, and so is the division by 2^wasted.  pipe_eval_cand_w32 / the packing kernel's wide form work on these.
FGI double pipe_cdbl(int32_t l, int32_t r, uint32_t C, bool ms, double wscale)
{
    const double dl = (double)l, dr = (double)r;
    double v;
    if (!ms) v = C == 0 ? dl : dr;
    else v = C == 0 ? dl : (C == 1 ? dr : (C == 2 ? __builtin_floor((dl + dr) * 0.5) : dl - dr));
    return v * wscale;
}

// ================================================================================================ staging: HBM -> LDS rows
// Cooperative over NT threads.  Rows of `seg` samples (+PADE of skew); g -> element index g + (g / seg) * PADE.
template <int NCH, bool ACC64, int NT, bool RAG>
FGI uint32_t pipe_stage(const void *pcm, const FgBlockDesc &d, const FgEncParams &P, LDS typename PipeTypes<ACC64>::samp_t *sL,
                        LDS typename PipeTypes<ACC64>::samp_t *sR, int tid, uint32_t seg, const PipeGeo &geo, uint32_t pre = 0)
{
    typedef typename PipeTypes<ACC64>::samp_t samp_t;
    constexpr uint32_t PADE = PipeTypes<ACC64>::PADE;
    const uint32_t n = d.n;
    const uint32_t magic = 0xFFFFFFFFu / (seg ? seg : 1) + 1;
    // (ragged geometry: exact quotients by S, base + 1 and base for indices below 2^16)
    const uint32_t mS = 0xFFFFFFFFu / geo.S + 1, mB1 = 0xFFFFFFFFu / (geo.base + 1) + 1, mB = 0xFFFFFFFFu / geo.base + 1;
    constexpr bool rag = RAG;
#define FGP_SADDR(g) (rag ? pipe_rag_addr(geo, (g), mS, mB1, mB) : ((g) + __umulhi((g), magic) * PADE))
    const int32_t lim = (int32_t)(P.bps - 1);
    uint32_t bad = 0;
    uint32_t istart = 0;
    if (NCH == 2 && !rag && !P.pcm_i16 && (d.pcm_off & 1) == 0 && (n & 127) == 0 && (seg & 1) == 0 && (((uintptr_t)pcm) & 15) == 0) {
        // two inter-channel samples per load; the pair lands in one LDS row (even index, even row length)
        const int4 *src = (const int4 *)((const int2 *)pcm + d.pcm_off);
        for (uint32_t j0 = 0; j0 < n / 2; j0 += 4 * NT) {
            int4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = j0 + u * NT + tid;
                v[u] = make_int4(0, 0, 0, 0);
                if (j < n / 2) v[u] = src[j];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = j0 + u * NT + tid;
                if (j < n / 2) {
                    const int32_t l0 = v[u].x, r0 = v[u].y, l1 = v[u].z, r1 = v[u].w;
                    if (P.bps < 32) bad |= (uint32_t)(((l0 ^ (l0 >> 31)) >> lim) | ((r0 ^ (r0 >> 31)) >> lim) | ((l1 ^ (l1 >> 31)) >> lim) | ((r1 ^ (r1 >> 31)) >> lim));
                    const uint32_t ad = FGP_SADDR(2 * j);
                    if (sizeof(samp_t) == 2) {
                        *(LDS uint32_t *)(sL + ad) = ((uint32_t)l0 & 0xFFFFu) | ((uint32_t)l1 << 16);
                        *(LDS uint32_t *)(sR + ad) = ((uint32_t)r0 & 0xFFFFu) | ((uint32_t)r1 << 16);
                    }
                    else { sL[ad] = (samp_t)(l0 >> pre); sL[ad + 1] = (samp_t)(l1 >> pre); sR[ad] = (samp_t)(r0 >> pre); sR[ad + 1] = (samp_t)(r1 >> pre); }
                }
            }
        }
        istart = n;
    }
    if (NCH == 2 && !rag && P.pcm_i16 && (d.pcm_off & 1) == 0 && (n & 127) == 0 && (seg & 1) == 0 && (((uintptr_t)pcm) & 7) == 0) {
        // int16 stereo: two inter-channel samples per 8-byte load, stored as one LDS word per channel
        const int2 *src = (const int2 *)((const short2 *)pcm + d.pcm_off);
        for (uint32_t j0 = 0; j0 < n / 2; j0 += 4 * NT) {
            int2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = j0 + u * NT + tid;
                v[u] = make_int2(0, 0);
                if (j < n / 2) v[u] = src[j];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = j0 + u * NT + tid;
                if (j < n / 2) {
                    const uint32_t w0 = (uint32_t)v[u].x, w1 = (uint32_t)v[u].y;     // (l0, r0), (l1, r1)
                    const uint32_t ad = FGP_SADDR(2 * j);
                    if (P.bps < 16) {
                        const int32_t l0 = (int16_t)w0, r0 = (int32_t)w0 >> 16, l1 = (int16_t)w1, r1 = (int32_t)w1 >> 16;
                        bad |= (uint32_t)(((l0 ^ (l0 >> 31)) >> lim) | ((r0 ^ (r0 >> 31)) >> lim) | ((l1 ^ (l1 >> 31)) >> lim) | ((r1 ^ (r1 >> 31)) >> lim));
                    }
                    if (sizeof(samp_t) == 2) {
                        *(LDS uint32_t *)(sL + ad) = (w0 & 0xFFFFu) | (w1 << 16);
                        *(LDS uint32_t *)(sR + ad) = (w0 >> 16) | (w1 & 0xFFFF0000u);
                    }
                    else { sL[ad] = (samp_t)(int16_t)w0; sL[ad + 1] = (samp_t)(int16_t)w1; sR[ad] = (samp_t)((int32_t)w0 >> 16); sR[ad + 1] = (samp_t)((int32_t)w1 >> 16); }
                }
            }
        }
        istart = n;
    }
    // any other layout: one sample per lane and load, four loads in flight (the input format is tested outside the loops)
    auto rest = [&](auto I16) __attribute__((always_inline)) {
        constexpr bool i16 = decltype(I16)::value;
        for (uint32_t i0 = istart; i0 < n; i0 += 4 * NT) {
            int32_t a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * NT + tid;
                a[u] = 0; b[u] = 0;
                if (i < n) {
                    if (NCH == 2) {
                        if (i16) { const short2 v = ((const short2 *)pcm)[d.pcm_off + i]; a[u] = v.x; b[u] = v.y; }
                        else { const int2 v = ((const int2 *)pcm)[d.pcm_off + i]; a[u] = v.x; b[u] = v.y; }
                    }
                    else {
                        const u64 at = d.pcm_off + (u64)i * (d.reserved ? d.reserved : 1u);
                        if (i16) a[u] = ((const int16_t *)pcm)[at];
                        else a[u] = ((const int32_t *)pcm)[at];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * NT + tid;
                if (i < n) {
                    if (P.bps < 32) bad |= (uint32_t)(((a[u] ^ (a[u] >> 31)) >> lim) | ((b[u] ^ (b[u] >> 31)) >> lim));
                    const uint32_t ad = FGP_SADDR(i);
                    sL[ad] = (samp_t)(a[u] >> pre);
                    if (NCH == 2) sR[ad] = (samp_t)(b[u] >> pre);
                }
            }
        }
    };
    if (istart < n) {
        if (P.pcm_i16) rest(std::integral_constant<bool, true>());
        else rest(std::integral_constant<bool, false>());
    }
#undef FGP_SADDR
    return bad;
}

// ================================================================================================ K4: evaluation of one candidate by one wave
// Rice parameter / partition order search over the 64 per-lane |residual| sums of one candidate (lane = partition of the
// finest order, or a power-of-two fraction of it).  At partition order po, partition p lives in lane p * (64 >> po); going
// one order down adds the neighbour 2^(5-po) lanes up.  Returns the best total in best_bits, its order in bpo, and leaves
// the parameter of partition p in lane p * (64 >> bpo) of kb.
template <bool ACC64>
FGI void pipe_rice_search(typename PipeTypes<ACC64>::sum_t psum, bool dead_in, int lane, uint32_t n, uint32_t order, uint32_t sb,
                          uint32_t pmin0, uint32_t pmax0, uint32_t limit, uint32_t &best_bits, uint32_t &bpo, uint32_t &kb)
{
    u64 sv = (u64)psum;
    best_bits = 0; bpo = 0; kb = 0;
    (void)dead_in;
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
        u64 o = up((uint32_t)sv, t);
        if (ACC64) o |= (u64)up((uint32_t)(sv >> 32), t) << 32;
        sv += o;
    };
    for (uint32_t m = 6; m > pmax0; m--) merge(6 - m);
    const uint32_t psz0 = n >> pmax0;
    {
        const bool wrap32 = (sb + 4) < (32 - ilog2_32(psz0));
        if (wrap32) sv &= 0xFFFFFFFFull;
    }
    // 0x40000 / x for x < 2^16 through the reciprocal, corrected to the exact quotient
    auto div18 = [&](uint32_t x) __attribute__((always_inline)) -> uint32_t {
        uint32_t qd = (uint32_t)(262144.0f * __builtin_amdgcn_rcpf((float)x));
        const int32_t r = (int32_t)(0x40000u - qd * x);
        if (r < 0) qd--;
        else if ((uint32_t)r >= x) qd++;
        return qd;
    };
    const bool lane0 = lane == 0;
    auto po_step = [&](auto PO) __attribute__((always_inline)) {
        constexpr int po = decltype(PO)::value;
        if (po > (int)pmax0 || po < (int)pmin0) return;
        constexpr uint32_t stride = 64u >> po;
        const bool valid = ((uint32_t)lane & (stride - 1)) == 0;
        const uint32_t pbase = n >> po;
        const uint32_t np = lane0 ? pbase - order : pbase;
        const uint32_t dv = lane0 ? div18(pbase - order) : div18(pbase);
        uint32_t kr, bits;
        if (!ACC64) {
            const uint32_t s32 = (uint32_t)sv;
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
            const u64 s = sv;
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
        if (best_bits == 0 || bits < best_bits) { best_bits = bits; bpo = (uint32_t)po; kb = kr; }
        if (po > (int)pmin0) merge(6 - (uint32_t)po);
    };
    po_step(std::integral_constant<int, 6>()); po_step(std::integral_constant<int, 5>());
    po_step(std::integral_constant<int, 4>()); po_step(std::integral_constant<int, 3>());
    po_step(std::integral_constant<int, 2>()); po_step(std::integral_constant<int, 1>());
    po_step(std::integral_constant<int, 0>());
}

// The candidate of a wave is a run-time value (one copy of the code for L, R, M and S instead of four -- the instruction
// cache is shared by the CU's waves): candidate sample = (ca * l + cb * r) >> (cs + wasted bits), with (ca, cb, cs) =
// L (1, 0, 0), R (0, 1, 0), M (1, 1, 1), S (1, -1, 0) in SGPRs: two 24-bit multiplies and one shift.
FGI void pipe_cand_coef(bool ms, uint32_t c, int32_t &ca, int32_t &cb, uint32_t &cs)
{
    if (!ms) { ca = c == 0 ? 1 : 0; cb = c == 0 ? 0 : 1; cs = 0; }
    else { ca = c == 1 ? 0 : 1; cb = c == 0 ? 0 : (c == 3 ? -1 : 1); cs = c == 2 ? 1 : 0; }
}
FGI int32_t pipe_cand(int32_t l, int32_t r, int32_t ca, int32_t cb, uint32_t sh)
{
    return (__mul24(l, ca) + __mul24(r, cb)) >> sh;
}

// The same search with every partition of every order evaluated at once.  The partition sums of all orders are differences
// of the inclusive prefix sums P of the 64 lane sums (partition p of order o = leaves [p << (6-o), (p+1) << (6-o))), so one
// scan and two cross-lane reads give every lane the sum of "its" partition: lanes 0..31 take the 32 partitions of order 5,
// lanes 32..47 order 4, 48..55 order 3, 56..59 order 2, 60..61 order 1, lane 62 order 0; order 6 (one partition per lane)
// is a round of its own.  One evaluation of (parameter, bits) per round instead of one per order, and a second scan gives the
// totals of all orders.  Results: best_bits / bpo as pipe_rice_search; the parameter of partition `kpart` in `kpar` on the
// lanes with kvalid.  Returns false (nothing computed) in the cases the sequential version handles: sums that wrapped in
// libFLAC's 32-bit accumulation, or totals that need 64-bit sums.
template <bool ACC64>
FGI bool pipe_rice_search_tree(typename PipeTypes<ACC64>::sum_t psum, int lane, uint32_t n, uint32_t order, uint32_t sb, uint32_t pmin0,
                               uint32_t pmax0, uint32_t limit, uint32_t &best_bits, uint32_t &bpo, uint32_t &kpar, uint32_t &kpart, bool &kvalid)
{
    // inclusive prefix sums of the lane sums
    u64 P;
    if (!ACC64) P = (u64)wave_scan_add((uint32_t)psum);          // (the block total of <= 16-bit input stays below 2^32)
    else {
        const u64 v = (u64)psum;
        const u64 a = wave_scan_add((uint32_t)v & 0xFFFF), b = wave_scan_add(((uint32_t)v >> 16) & 0xFFFF), c = wave_scan_add((uint32_t)(v >> 32));
        P = a + (b << 16) + (c << 32);
    }
    if (ACC64) {
        // libFLAC sums a partition in 32 bits when bps + 4 + log2(partition size) < 32: harmless unless the residual is wilder
        // than that bound, which the sequential version reproduces
        const uint32_t tot_hi = rl((uint32_t)(P >> 32), 63);
        const bool wrap32 = (sb + 4) < (32 - ilog2_32(n >> pmax0));
        if (wrap32 && tot_hi != 0) return false;
    }
    auto div18 = [&](uint32_t x) __attribute__((always_inline)) -> uint32_t {
        uint32_t qd = (uint32_t)(262144.0f * __builtin_amdgcn_rcpf((float)x));
        const int32_t r = (int32_t)(0x40000u - qd * x);
        if (r < 0) qd--;
        else if ((uint32_t)r >= x) qd++;
        return qd;
    };
    // (parameter, bits) of one partition: sum s, np samples
    auto node = [&](u64 s, uint32_t np, uint32_t &kr, uint32_t &pb) __attribute__((always_inline)) {
        const uint32_t dv = div18(np);
        if (!ACC64) {
            const uint32_t s32 = (uint32_t)s;
            const uint32_t s1 = (s32 > 1 ? s32 : 1) - 1;
            const uint32_t qv = (uint32_t)(((u64)s1 * dv) >> 18);
            kr = qv ? 32 - (uint32_t)__builtin_clz(qv) : 0;
            if (kr >= limit) kr = limit - 1;
            pb = 4 + (1 + kr) * np + ((s32 << 1) >> kr) - (np >> 1);
        }
        else {
            kr = 0;
            if (s >= 2) {
                const u64 qv = ((s - 1) * dv) >> 18;
                if (qv != 0) kr = ilog2_64(qv) + 1;
            }
            if (kr >= limit) kr = limit - 1;
            u64 b = (u64)4 + (u64)(1 + kr) * np + (kr ? (s >> (kr - 1)) : (s << 1)) - (np >> 1);
            if (b > 0xFFFFFFFFull) b = 0xFFFFFFFFull;
            pb = (uint32_t)b;
        }
    };
    // ---- orders 5..0: lane -> (order lev, partition p)
    const uint32_t lo = (uint32_t)__builtin_clz(~((uint32_t)lane << 26));     // leading ones of the 6-bit lane number
    const int lev = 5 - (int)lo;                                               // -1 for lane 63
    const uint32_t base = 64u - (64u >> lo);
    const uint32_t p = (uint32_t)lane - base;
    const bool inB = lev >= 0 && (uint32_t)lev <= pmax0 && (uint32_t)lev >= pmin0;
    const uint32_t width = 64u >> (lev < 0 ? 0 : lev);                         // leaves per partition
    const uint32_t first = p * width, last = first + width - 1;                // leaf range
    u64 hi_ = (u64)(uint32_t)__shfl((int)(uint32_t)P, (int)(last & 63));
    u64 lo_ = (u64)(uint32_t)__shfl((int)(uint32_t)P, (int)((first - 1) & 63));
    if (ACC64) {
        hi_ |= (u64)(uint32_t)__shfl((int)(uint32_t)(P >> 32), (int)(last & 63)) << 32;
        lo_ |= (u64)(uint32_t)__shfl((int)(uint32_t)(P >> 32), (int)((first - 1) & 63)) << 32;
    }
    const u64 sB = hi_ - (first == 0 ? 0 : lo_);
    uint32_t kB, pbB;
    {
        const uint32_t pbase = n >> (lev < 0 ? 0 : lev);
        node(sB, (p == 0) ? pbase - order : pbase, kB, pbB);
        if (!inB) pbB = 0;
    }
    // ---- order 6: one partition per lane
    uint32_t kA = 0, pbA = 0;
    const bool doA = pmax0 == 6;
    if (doA) node((u64)psum, (lane == 0) ? (n >> 6) - order : (n >> 6), kA, pbA);
    if (__any((pbA | pbB) >> 25)) return false;                                // totals beyond 32 bits: the sequential version
    const uint32_t SB = wave_scan_add(pbB);
    const uint32_t t5 = rl(SB, 31), t4 = rl(SB, 47), t3 = rl(SB, 55), t2 = rl(SB, 59), t1 = rl(SB, 61), t0 = rl(SB, 62);
    const uint32_t tot[7] = {t0 - t1, t1 - t2, t2 - t3, t3 - t4, t4 - t5, t5, doA ? wave_sum(pbA) : 0u};
    best_bits = 0; bpo = 0;
#pragma unroll
    for (int po = 6; po >= 0; po--) {
        if ((uint32_t)po > pmax0 || (uint32_t)po < pmin0) continue;
        const uint32_t bits = tot[po] + 6;
        if (best_bits == 0 || bits < best_bits) { best_bits = bits; bpo = (uint32_t)po; }
    }
    if (bpo == 6) { kpar = kA; kpart = (uint32_t)lane; kvalid = true; }
    else { kpar = kB; kpart = p; kvalid = lev == (int)bpo; }
    return true;
}

template <bool MS, int NCH, int MAXO, bool ACC64, bool RAG>
FGI void pipe_eval_cand(const uint32_t C, const FgBlockDesc &d, uint32_t bi, const FgEncParams &P, const FgPipeBufs &B, FgBlockResult *results,
                        FgDebugRec *mydbg, const LDS typename PipeTypes<ACC64>::samp_t *sL, const LDS typename PipeTypes<ACC64>::samp_t *sR,
                        int lane, uint32_t range_err, const PipeGeo &geo, uint32_t pre = 0, bool pre_ok = true)
{
    constexpr int NC = MS ? 4 : NCH;
    typedef typename PipeTypes<ACC64>::sum_t sum_t;
    typedef typename PipeTypes<ACC64>::samp_t samp_t;
    constexpr uint32_t PADE = PipeTypes<ACC64>::PADE;
    const uint32_t n = d.n;
    // `seg`: the samples every working lane has (RAG: `base`; the first `extra` lanes of a group have one more -- ln.len)
    const uint32_t seg = RAG ? geo.base : n >> 6, rstr = RAG ? geo.rstr : seg + PADE;
    (void)PADE;
    const PipeLane ln = pipe_lane<RAG>(geo, (uint32_t)lane, seg);
    const LDS samp_t *rowL = sL + (uint32_t)lane * rstr, *rowR = sR + (uint32_t)lane * rstr;
    const LDS samp_t *prvL = sL + ln.prow * rstr + ln.plen, *prvR = sR + ln.prow * rstr + ln.plen;      // one past the samples in front
    const uint32_t wraw = rfl(B.wasted[bi * NC + C]);
    const uint32_t wst = wraw & 0xFFu;
    const uint32_t nominal = P.bps + ((MS && C == 3) ? 1u : 0u);
    const uint32_t sb = nominal - wst;
    // (32-bit streams, pipe_preshift: a block the staged forms cannot hold goes to the generic kernel -- bit 31 of the decision's
    // size tells the packing kernel)
    const bool unsupported = ACC64 && P.bps == 32 && (!pre_ok || (sb > 25 && !(wraw & 0x100u)));
    if (unsupported) {
        if (lane == 0) {
            FgPipeDec *dec = B.dec + (size_t)bi * NC + C;
            dec->bits = 0x80000000u; dec->type = 1; dec->order = 0; dec->prec = 0; dec->shift = 0; dec->porder = 0; dec->method = 0; dec->wasted = wraw;
            FgBlockResult *r = &results[d.out_slot];
            r->best_bits[C] = 0;
            if (C == 0) {
                r->bytes = 0; r->ca = 0; r->err = range_err; r->reserved = 4;
#pragma unroll
                for (int c = NC; c < 4; c++) r->best_bits[c] = 0;
                for (int w = 0; w < 4; w++) B.chunk_bits[(size_t)d.out_slot * 4 + w] = 0;
            }
        }
        return;
    }
    // candidate value of sample s of this lane's row (row offset `ro` = 0 or -rstr for the left neighbour)
    // Round 5: candidate = (a + cb b) >> shift with the channels picked per candidate -- L: a = left, cb = 0; R: a = right, cb = 0;
    // M: a = left, b = right, cb = 1, shift 1; S: cb = -1 -- one multiply-add and one shift where (ca l + cb r) >> shift took a select more.
    int32_t cca, ccb;
    uint32_t ccs;
    pipe_cand_coef(MS, C, cca, ccb, ccs);
    (void)cca;
    const bool cplain = !MS || C < 2;
    const int32_t cvb = cplain ? 0 : ccb;
    const LDS samp_t *const baseA = (NCH == 2 && C == 1) ? sR : sL, *const baseB = cplain ? baseA : sR;
    const LDS samp_t *const rowA = baseA + (uint32_t)lane * rstr, *const rowB = baseB + (uint32_t)lane * rstr;
    const LDS samp_t *const prvA = baseA + ln.prow * rstr + ln.plen, *const prvB = baseB + ln.prow * rstr + ln.plen;
    const uint32_t csh = (wraw & 0x100u) ? ccs : ccs + wst - pre;       // (staged samples are already down by `pre`)
    auto cv = [&](int32_t a, int32_t b) __attribute__((always_inline)) -> int32_t { return (a + __mul24(b, cvb)) >> csh; };
    auto samp = [&](int s) __attribute__((always_inline)) -> int32_t { return cv(rowA[s], rowB[s]); };
    // the k-th sample in front of this lane's first one (k >= 1)
    auto hsamp = [&](int k) __attribute__((always_inline)) -> int32_t {
        if (!RAG) return samp((int)seg - k - (int)rstr);
        return cv(prvA[-k], prvB[-k]);
    };

    uint32_t pmax0 = 0;
    { uint32_t b = n; while (!(b & 1)) { pmax0++; b >>= 1; } if (pmax0 > 15) pmax0 = 15; }
    if (P.max_po < pmax0) pmax0 = P.max_po;
    const uint32_t pmin0 = P.min_po < pmax0 ? P.min_po : pmax0;

    // ================================================================ fixed-predictor error sums (one pass)
    // facc = sums over the lane's samples from sample 4 of the block on (what libFLAC's order guess uses), fwarm = the part
    // of samples 0..3 that belongs to the order-k residual (s >= k): the per-lane sums double as the Rice partition sums of
    // the fixed predictors.
    u64 tot[5];
    sum_t fsum = 0;            // per-lane sum of |residual| of the fixed predictor of the guessed order (set below)
    uint32_t guess;
    // Round 5 (<= 16 bit, regular geometry): ONE walk over the lane's samples serves the fixed-predictor sums and the residual of the
    // first LPC vector -- its quantised coefficients are there when the kernel starts --, so the candidate's samples are formed once
    // where two walks formed them twice.  m_*: that vector's facts and the lane's sum of |residual|, picked up by pass 1 below.
    bool m_have = false;
    uint32_t m_order = 0, m_prec = 0;
    int m_shift = 0;
    int32_t m_q[MAXO];
    sum_t m_psum = 0;
#pragma unroll
    for (int j = 0; j < MAXO; j++) m_q[j] = 0;
    if constexpr (!ACC64 && !RAG) {
        if (FGX_MERGE && P.max_lpc_order > 0 && P.nvec > 0 && seg >= (uint32_t)MAXO && rfl(B.nv[bi]) > 0) {
            const size_t ridx = ((size_t)bi * NC + C) * P.nvec;
            const uint32_t r = rfl(B.lres[ridx]);
            if ((r >> 24) & 1) {
                m_have = true;
                m_order = r & 0xFF; m_prec = (r >> 8) & 0xFF; m_shift = (int)(int8_t)((r >> 16) & 0xFF);
                if (m_order == 0) m_order = 1;
                const int32_t qall = (lane < MAXO) ? B.qres[ridx * MAXO + lane] : 0;
#pragma unroll
                for (int j = 0; j < MAXO; j++) m_q[j] = (int32_t)rl((uint32_t)qall, j);
            }
        }
    }
    {
        sum_t facc[5], fwarm[5];
        // Everything carries a bias FB (values and differences stay far below it), so |a - b| is one v_sad_u32 on the biased
        // values -- with the running sum as its addend for <= 16-bit input -- and the differences themselves
        // (e_k = e_{k-1} - previous e_{k-1}) keep the bias with one extra add: 12 instructions per sample instead of 19.
        constexpr uint32_t FB = 1u << 30;
        uint32_t P0 = FB, P1 = FB, P2 = FB, P3 = FB;      // previous value and previous 1st..3rd differences (of zeros)
#pragma unroll
        for (int kk = 0; kk < 5; kk++) { facc[kk] = 0; fwarm[kk] = 0; }
        bool merged_walk = false;
        if constexpr (!ACC64 && !RAG) {
            if (m_have) {
                merged_walk = true;
                // the MAXO samples in front of the lane's first one (zeros for lane 0): history of the filter, and of the differences
                int32_t h[MAXO];
#pragma unroll
                for (int j = 0; j < MAXO; j++) {
                    int32_t x = 0;
                    if (lane > 0) x = hsamp(1 + j);
                    h[(MAXO - 1 - j) % MAXO] = x;
                }
#pragma unroll
                for (int k = 4; k >= 1; k--) {
                    const uint32_t vb = (uint32_t)h[(MAXO - k) % MAXO] + FB;
                    const uint32_t e1b = vb - P0 + FB, e2b = e1b - P1 + FB, e3b = e2b - P2 + FB;
                    P0 = vb; P1 = e1b; P2 = e2b; P3 = e3b;
                }
                uint32_t psum = 0;
                auto mstep = [&](int u, uint32_t s, auto PRO, bool guard) __attribute__((always_inline)) {
                    constexpr bool pro = decltype(PRO)::value;      // samples 0..3 of the lane: the sums tell warm-up positions apart
                    const int32_t x = samp((int)s);
                    const int32_t res = x - (pfir24<MAXO>(m_q, h, u) >> m_shift);
                    h[u] = x;
                    const bool real = !guard || lane > 0 || s >= m_order;
                    if (real) psum += pabs32(res);
                    const uint32_t vb = (uint32_t)x + FB;
                    const uint32_t e1b = vb - P0 + FB, e2b = e1b - P1 + FB, e3b = e2b - P2 + FB;
                    if (pro) {
                        const uint32_t ab[5] = {psad(vb, FB, 0), psad(vb, P0, 0), psad(e1b, P1, 0), psad(e2b, P2, 0), psad(e3b, P3, 0)};
#pragma unroll
                        for (int kk = 0; kk < 5; kk++) {
                            facc[kk] += (lane > 0) ? ab[kk] : 0u;
                            if ((int)s >= kk) fwarm[kk] += ab[kk];
                        }
                    }
                    else {
                        facc[0] = (sum_t)psad(vb, FB, (uint32_t)facc[0]); facc[1] = (sum_t)psad(vb, P0, (uint32_t)facc[1]);
                        facc[2] = (sum_t)psad(e1b, P1, (uint32_t)facc[2]); facc[3] = (sum_t)psad(e2b, P2, (uint32_t)facc[3]);
                        facc[4] = (sum_t)psad(e3b, P3, (uint32_t)facc[4]);
                    }
                    P0 = vb; P1 = e1b; P2 = e2b; P3 = e3b;
                };
                typedef std::integral_constant<bool, true> TT;
                typedef std::integral_constant<bool, false> FF;
#pragma unroll
                for (int u = 0; u < MAXO; u++) { if (u < 4) mstep(u, (uint32_t)u, TT(), true); else mstep(u, (uint32_t)u, FF(), true); }
                uint32_t s0 = MAXO;
#pragma unroll 1
                for (; s0 + MAXO <= seg; s0 += MAXO) {
#pragma unroll
                    for (int u = 0; u < MAXO; u++) mstep(u, s0 + u, FF(), false);
                }
#pragma unroll
                for (int u = 0; u < MAXO; u++) if (s0 + u < seg) mstep(u, s0 + u, FF(), false);
                m_psum = (sum_t)psum;
            }
        }
        if (!merged_walk) {
#pragma unroll
        for (int s = -4; s < 4; s++) {
            int32_t v = 0;
            if (s >= 0) v = samp(s);
            else if (lane > 0) v = hsamp(-s);
            const uint32_t vb = (uint32_t)v + FB;
            const uint32_t e1b = vb - P0 + FB, e2b = e1b - P1 + FB, e3b = e2b - P2 + FB;
            if (s >= 0) {
                const uint32_t ab[5] = {psad(vb, FB, 0), psad(vb, P0, 0), psad(e1b, P1, 0), psad(e2b, P2, 0), psad(e3b, P3, 0)};
#pragma unroll
                for (int kk = 0; kk < 5; kk++) {
                    facc[kk] += (lane > 0) ? ab[kk] : 0u;
                    if (s >= kk) fwarm[kk] += ab[kk];
                }
            }
            P0 = vb; P1 = e1b; P2 = e2b; P3 = e3b;
        }
        }
        auto fstep = [&](int s) __attribute__((always_inline)) {
            const uint32_t vb = (uint32_t)samp(s) + FB;
            const uint32_t e1b = vb - P0 + FB, e2b = e1b - P1 + FB, e3b = e2b - P2 + FB;
            if (!ACC64) {
                facc[0] = (sum_t)psad(vb, FB, (uint32_t)facc[0]); facc[1] = (sum_t)psad(vb, P0, (uint32_t)facc[1]);
                facc[2] = (sum_t)psad(e1b, P1, (uint32_t)facc[2]); facc[3] = (sum_t)psad(e2b, P2, (uint32_t)facc[3]);
                facc[4] = (sum_t)psad(e3b, P3, (uint32_t)facc[4]);
            }
            else {
                facc[0] += psad(vb, FB, 0); facc[1] += psad(vb, P0, 0); facc[2] += psad(e1b, P1, 0);
                facc[3] += psad(e2b, P2, 0); facc[4] += psad(e3b, P3, 0);
            }
            P0 = vb; P1 = e1b; P2 = e2b; P3 = e3b;
        };
        int s4 = 4;
        if (merged_walk) s4 = (int)seg;
#pragma unroll 1
        for (; s4 + 4 <= (int)seg; s4 += 4) {
#pragma unroll
            for (int t = 0; t < 4; t++) fstep(s4 + t);
        }
#pragma unroll 1
        for (; s4 < (int)seg; s4++) fstep(s4);
        if (RAG) {
            // the one sample more that the first lanes of a group have; idle lanes have walked over whatever row 0 holds
            sum_t keep[5];
#pragma unroll
            for (int kk = 0; kk < 5; kk++) keep[kk] = facc[kk];
            fstep((int)seg);
#pragma unroll
            for (int kk = 0; kk < 5; kk++) { if (ln.len <= seg) facc[kk] = keep[kk]; if (!ln.act) { facc[kk] = 0; fwarm[kk] = 0; } }
        }
#pragma unroll
        for (int kk = 0; kk < 5; kk++) tot[kk] = ACC64 ? wave_sum64((u64)facc[kk]) : (u64)wave_sum((uint32_t)facc[kk]);
        if constexpr (RAG) {
            // The reference binary sums these errors with its AVX2 routine whenever 32-bit accumulators might overflow
            // (sb + ilog2((n - 4) * 17) >= 32, stream_encoder.c process_subframe_): four lanes of q = len / 4 samples that start
            // at (j len) / 4 but take their history from j q.  With len % 4 != 0 lanes 2 and 3 start one or two samples late
            // against their history, a sample or two between the lanes and the len % 4 samples at the end are not counted
            // (oracle/flac_oracle.c avx2_lane_sums has the derivation; regular blocks have len % 4 == 0).  The sums above are
            // the exact ones: take out what the routine skips, and swap the first four errors of a shifted lane for what it
            // computes -- a dozen samples, done by every lane alike.
            const uint32_t len = n - 4, qq = len >> 2, rr = len & 3;
            if (rr != 0 && sb + ilog2_32(len * 17) >= 32) {
                const uint32_t mS = 0xFFFFFFFFu / geo.S + 1, mB1 = 0xFFFFFFFFu / (geo.base + 1) + 1, mB = 0xFFFFFFFFu / geo.base + 1;
                auto xat = [&](uint32_t g) -> i64 {          // candidate value of sample g of the block
                    const uint32_t ad = pipe_rag_addr(geo, g, mS, mB1, mB);
                    return (i64)cv(baseA[ad], baseB[ad]);
                };
                auto aabs = [](i64 v) -> u64 { return (u64)(v < 0 ? -v : v); };
                // exact errors of orders 0..4 at d[i] (= sample i + 4 of the block)
                auto exact = [&](uint32_t i, u64 (&e)[5]) {
                    const i64 a = xat(i + 4), b = xat(i + 3), c = xat(i + 2), dd = xat(i + 1), ee = xat(i);
                    e[0] = aabs(a); e[1] = aabs(a - b); e[2] = aabs(a - 2 * b + c); e[3] = aabs(a - 3 * b + 3 * c - dd);
                    e[4] = aabs(a - 4 * b + 6 * c - 4 * dd + ee);
                };
                i64 adj[5] = {0, 0, 0, 0, 0};
                const uint32_t st2 = len >> 1, st3 = (3 * len) >> 2;
                auto skip = [&](uint32_t lo, uint32_t hi) {
                    for (uint32_t i = lo; i < hi; i++) { u64 e[5]; exact(i, e); for (int k = 0; k < 5; k++) adj[k] -= (i64)e[k]; }
                };
                skip(2 * qq, st2); skip(st2 + qq, st3); skip(st3 + qq, len);
                auto shifted = [&](uint32_t j, uint32_t st) {
                    const uint32_t hb = j * qq;
                    if (st == hb) return;
                    // state from the history at hb, data from st
                    const i64 h1 = xat(hb + 3), h2 = xat(hb + 2), h3 = xat(hb + 1), h4 = xat(hb);
                    i64 p0 = h1, p1 = h1 - h2, p2 = p1 - (h2 - h3), p3 = p2 - (h2 - 2 * h3 + h4);
                    for (uint32_t i = 0; i < 4 && i < qq; i++) {
                        const i64 e0 = xat(st + i + 4), e1 = e0 - p0, e2 = e1 - p1, e3 = e2 - p2, e4 = e3 - p3;
                        u64 ex[5];
                        exact(st + i, ex);
                        adj[0] += (i64)aabs(e0) - (i64)ex[0]; adj[1] += (i64)aabs(e1) - (i64)ex[1]; adj[2] += (i64)aabs(e2) - (i64)ex[2];
                        adj[3] += (i64)aabs(e3) - (i64)ex[3]; adj[4] += (i64)aabs(e4) - (i64)ex[4];
                        p3 = e3; p2 = e2; p1 = e1; p0 = e0;
                    }
                };
                shifted(2, st2); shifted(3, st3);
#pragma unroll
                for (int kk = 0; kk < 5; kk++) tot[kk] = (u64)((i64)tot[kk] + adj[kk]);
            }
        }
        // fixed order guess (fixed.c: the smallest total error wins, lower order on ties)
        const u64 m34 = tot[3] < tot[4] ? tot[3] : tot[4];
        const u64 m234 = tot[2] < m34 ? tot[2] : m34;
        const u64 m1234 = tot[1] < m234 ? tot[1] : m234;
        if (tot[0] <= m1234) guess = 0;
        else if (tot[1] <= m234) guess = 1;
        else if (tot[2] <= m34) guess = 2;
        else if (tot[3] <= tot[4]) guess = 3;
        else guess = 4;
        // the lane's partition sum of that predictor: lane 0 adds the part of samples g..3
        sum_t a = guess == 0 ? facc[0] : guess == 1 ? facc[1] : guess == 2 ? facc[2] : guess == 3 ? facc[3] : facc[4];
        const sum_t w = guess == 0 ? fwarm[0] : guess == 1 ? fwarm[1] : guess == 2 ? fwarm[2] : guess == 3 ? fwarm[3] : fwarm[4];
        if (lane == 0) a += w;
        fsum = a;
    }

    // ---- baseline: verbatim / constant
    uint32_t best;
    uint32_t d_type = 1, d_order = 0, d_prec = 0, d_porder = 0, d_method = 0, d_k = 0, d_kpart = 0;
    bool d_kvalid = false;      // this lane holds the Rice parameter d_k of partition d_kpart
    int d_shift = 0;
    int32_t bestq[MAXO];
#pragma unroll
    for (int j = 0; j < MAXO; j++) bestq[j] = 0;
    bool do_fixed = false, do_lpc = false;
    {
        const u64 vb = (u64)8 + (u64)n * sb;
        best = vb < 0xFFFFFFFFull ? (uint32_t)vb : 0xFFFFFFFFu;
        const uint32_t g = guess;
        const u64 tg = g == 0 ? tot[0] : g == 1 ? tot[1] : g == 2 ? tot[2] : g == 3 ? tot[3] : tot[4];
        const double len = (double)(n - 4);
        // libFLAC: bits per sample of the guessed fixed predictor, rbg = (float)(log(ln2 * total / len) / ln2); the fixed predictor is
        // evaluated unless rbg >= bits per sample.  Far below 2^(sb - 1) the answer needs no logarithm (its two hundred instructions
        // ran on every wave): ln2 total < len 2^(sb - 2) leaves log2 of the ratio more than a bit under sb - 1.
        bool fixed_worth;
        {
            const double num = FG_LN2 * (double)tg;
            if (tg > 0 && sb >= 3 && num < len * __hiloint2double((int)((1023u + sb - 2u) << 20), 0)) fixed_worth = true;
            else {
                const float rbg = (float)((tg > 0) ? log(num / len) / FG_LN2 : 0.0);
                fixed_worth = !(rbg >= (float)sb);
            }
        }
        bool constant = false;
        if (tot[1] == 0) {
            const int32_t x0 = cv(baseA[0], baseB[0]);
            uint32_t ne = 0;
#pragma unroll 1
            for (uint32_t s = 0; s < seg + (RAG ? 1u : 0u); s++) ne |= ((!RAG || s < ln.len) && samp((int)s) != x0);
            constant = !__any(ne != 0);
        }
        if (mydbg && lane == 0) {
            for (int kk = 0; kk < 5; kk++) mydbg->cand[C].fixed_tot[kk] = tot[kk];
            mydbg->cand[C].fixed_guess = g;
        }
        // limit_min_bitrate (libFLAC 1.4.3 as observed, oracle/flac_oracle.c): the last independent channel is evaluated with
        // CONSTANT disabled when every earlier one chose CONSTANT; mid and side of the same frame then are, too -- unless only
        // mid/side are evaluated at all (loose mid-side follower frames, forced_ca == 3).  "Every earlier independent channel
        // chose CONSTANT" concerns the left channel only here (NCH <= 2): each wave looks at it itself.
        if (P.limit_min_bitrate && C >= (uint32_t)(NCH - 1) && d.forced_ca != 3) {
            bool forbid = true;
            if (NCH == 2) {
                // is the left channel constant?  (its wasted bits do not matter: all samples equal either way)
                const int32_t l0 = sL[0];
                uint32_t ne = 0;
#pragma unroll 1
                for (uint32_t s = 0; s < seg + (RAG ? 1u : 0u); s++) ne |= ((!RAG || s < ln.len) && (int32_t)rowL[s] != l0);
                forbid = !__any(ne != 0);
                // ("chose CONSTANT", not "is constant": from 28 bits per sample on libFLAC's order guess -- the _limit_residual
                // forms -- flags only an all-zero signal as constant, oracle fixed_best_predictor)
                const uint32_t wl = rfl(B.wasted[(size_t)bi * NC]);
                if (forbid && !(wl & 0x100u) && P.bps - (wl & 0xFFu) >= 28) forbid = false;
            }
            if (forbid) constant = false;
        }
        if (constant) {
            const uint32_t cb = 8 + sb;
            if (cb < best) { best = cb; d_type = 0; }
        }
        else {
            if (fixed_worth) do_fixed = true;
            if (P.max_lpc_order > 0) do_lpc = true;
        }
    }
    const uint32_t nv = do_lpc ? rfl(B.nv[bi]) : 0;
    const uint32_t limit = P.rice_limit;

    // ================================================================ pass 0 = fixed predictor of the guessed order, then one
    // pass per autocorrelation vector
#pragma unroll 1
    for (uint32_t pass = 0; pass < 1 + nv; pass++) {
        uint32_t order, prec = 0;
        int shift = 0;
        int32_t q[MAXO];
        sum_t psum;
        uint32_t ovf = 0;
        double psumd = 0.0, pmaxd = 0.0;        // (fp64 forms: the lane's sum and maximum of |residual|; the sum is exact below 2^53)
        const int kind = pass == 0 ? 0 : 1;
        if (pass == 0) {
            if (!do_fixed) continue;
            order = guess;
            psum = fsum;
        }
        else {
            const uint32_t v = pass - 1;
            const size_t ridx = ((size_t)bi * NC + C) * P.nvec + v;
            const uint32_t r = rfl(B.lres[ridx]);
            order = r & 0xFF; prec = (r >> 8) & 0xFF; shift = (int)(int8_t)((r >> 16) & 0xFF);
            if (mydbg && lane == 0) mydbg->cand[C].lpc_guess[v] = ((r >> 25) & 1) ? (r & 0xFF) : 0;
            if (!((r >> 24) & 1)) continue;
            if (order == 0) order = 1;
            const bool from_merged = !ACC64 && !RAG && pass == 1 && m_have;      // (the walk above has this vector's sums)
            if (from_merged) {
#pragma unroll
                for (int j = 0; j < MAXO; j++) q[j] = m_q[j];
                psum = m_psum;
            }
            else {
            const int32_t qall = (lane < MAXO) ? B.qres[ridx * MAXO + lane] : 0;
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = (int32_t)rl((uint32_t)qall, j);
            // ---- FIR over the segment: history in registers, statically indexed (the loop is unrolled by its length)
            int32_t h[MAXO];
            double hd[MAXO], qd[MAXO];          // (17..25-bit samples: history and coefficients as doubles, pfir_f64n)
            const double scl = __hiloint2double((int)((1023u - (uint32_t)shift) << 20), 0);      // 2^-shift
            if constexpr (FGP_F64 && ACC64) {
                const double qdl = (double)qall * scl;         // (scaled in the lane that holds it: what is read out of it stays in SGPRs)
#pragma unroll
                for (int j = 0; j < MAXO; j++) qd[j] = __hiloint2double((int)rl((uint32_t)__double2hiint(qdl), j), (int)rl((uint32_t)__double2loint(qdl), j));
            }
            // (fp64 forms: the walk runs WITHOUT the running maximum first -- a lane whose sum of |residual| stays below 2^31 holds no
            // residual that reaches it, and that is every lane of every block of ordinary material; only when a sum gets there the walk is
            // repeated with the maximum: one instruction of eighteen a sample and window)
            auto walk = [&](auto TRACK) __attribute__((always_inline)) {
            psum = 0; psumd = 0.0; pmaxd = 0.0;
#pragma unroll
            for (int j = 0; j < MAXO; j++) {
                int32_t x = 0;
                if (lane > 0) x = hsamp(1 + j);
                h[(MAXO - 1 - j) % MAXO] = ACC64 ? ppack(x) : x;
                if constexpr (FGP_F64 && ACC64) hd[(MAXO - 1 - j) % MAXO] = (double)x;
            }
            auto step = [&](int u, uint32_t s, bool guard) __attribute__((always_inline)) {
                const int32_t x = samp((int)s);
                // warm-up samples (the first `order` of the block, all in lane 0) are not residuals: they neither count nor can
                // they overflow (a predictor with large coefficients -- a square wave under a punched window -- sends the
                // 'residual' of a sample whose history is still zeros far beyond 32 bits; libFLAC never forms it)
                const bool real = !guard || lane > 0 || s >= order;
                int32_t res;
                if constexpr (!ACC64) res = x - (pfir24<MAXO>(q, h, u) >> shift);
                else if constexpr (FGP_F64) {
                    // x - (sum >> shift): the scaling by 2^-shift is exact, floor() is the arithmetic shift
                    // The lane's sum and maximum of |residual| stay in fp64 -- v_add_f64 and v_max_f64 with the |x| modifier instead of
                    // a conversion, an absolute value, a 64-bit integer add and the two-sided range test of every residual; a residual
                    // is outside the 32-bit range exactly when its magnitude reaches 2^31 (libFLAC's _limit_residual test)
                    // (qd carries the scaling; the chain starts at -x: what comes out of the floor is minus the residual, pfir_f64n)
                    const double xd = (double)x;
                    const double rr = __builtin_floor(pfir_f64n<MAXO>(qd, hd, u, -xd));
                    if (real) { psumd += __builtin_fabs(rr); if (decltype(TRACK)::value) pmaxd = __builtin_fmax(pmaxd, __builtin_fabs(rr)); }
                    res = 0;
                    hd[u] = xd;
                }
                else {
                    const i64 rr = (i64)x - (pfir48<MAXO>(q, h, u) >> shift);
                    if ((rr <= (i64)INT32_MIN || rr > (i64)INT32_MAX) && real) ovf = 1;
                    res = (int32_t)rr;
                }
                h[u] = ACC64 ? ppack(x) : x;
                if constexpr (!(FGP_F64 && ACC64)) { if (real) psum += pabs32(res); }
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
            if (!RAG) {
#pragma unroll
                for (int u = 0; u < MAXO; u++) if (s0 + u < seg) step(u, s0 + u, s0 == 0);
            }
            else {
                // the rest of the lane (the common `seg` samples and the one more of the first lanes of a group); sample
                // s0 + u sits in history slot u
                // (at most MAXO of them: seg - s0 < MAXO, plus one)
#pragma unroll
                for (int u = 0; u < MAXO; u++) if (s0 + u < ln.len) step(u, s0 + u, s0 == 0);
                if (!ln.act) { psum = 0; ovf = 0; psumd = 0.0; pmaxd = 0.0; }
            }
            };
            if constexpr (FGP_F64 && ACC64) {
                walk(std::false_type());
                if (__any(!(psumd < 2147483648.0))) walk(std::true_type());
            }
            else walk(std::true_type());
            if constexpr (FGP_F64 && ACC64) {
                if (!(pmaxd < 2147483648.0)) ovf = 1;
                psum = ovf ? (sum_t)0 : (sum_t)psumd;
            }
            }
        }
        const bool dead = ACC64 && __any(ovf != 0);
        uint32_t best_bits, bpo, kb;
        uint32_t kpart;
        bool kvalid;
        // (format.c FLAC__format_get_max_rice_partition_order_from_blocksize_limited_max_and_predictor_order: a partition must
        // be longer than the predictor order -- never binding with 16 or more samples in the finest partition, i.e. for
        // regular blocks; short blocks of the ragged geometry meet it)
        uint32_t pmax_p = pmax0;
        if (RAG) while (pmax_p > 0 && (n >> pmax_p) <= order) pmax_p--;
        const uint32_t pmin_p = pmin0 < pmax_p ? pmin0 : pmax_p;
        if (!pipe_rice_search_tree<ACC64>(psum, lane, n, order, sb, pmin_p, pmax_p, limit, best_bits, bpo, kb, kpart, kvalid)) {
            pipe_rice_search<ACC64>(psum, dead, lane, n, order, sb, pmin_p, pmax_p, limit, best_bits, bpo, kb);
            kvalid = ((uint32_t)lane & ((64u >> bpo) - 1)) == 0;
            kpart = (uint32_t)lane >> (6 - bpo);
        }
        uint32_t est = 0;
        if (!dead) {
            est = kind == 0 ? (8 + order * sb) : (8 + 4 + 5 + order * (prec + sb));
            if (best_bits < 0xFFFFFFFFu - est) est += best_bits; else est = 0xFFFFFFFFu;
            if (est > 0 && est < best) {
                best = est;
                d_type = kind == 0 ? 2 : 3; d_order = order; d_prec = prec; d_shift = shift;
                d_porder = bpo; d_k = kb; d_kpart = kpart; d_kvalid = kvalid;
                d_method = __any(kvalid && kb >= 15) ? 1 : 0;
                if (kind == 1) {
#pragma unroll
                    for (int j = 0; j < MAXO; j++) bestq[j] = q[j];
                }
            }
        }
        if (mydbg && lane == 0) {
            // (the records carry libFLAC's numbers: with the wasted-bits field, see the decision record below)
            const uint32_t estw = (est == 0 || est > 0xFFFFFFFFu - wst) ? est : est + wst;
            if (kind == 0) mydbg->cand[C].fixed_bits = estw;
            else mydbg->cand[C].lpc_bits[pass - 1] = estw;
        }
    }

    // ---- the decision record of this candidate
    // Every estimate above left out the unary wasted-bits field of the subframe header (`wasted` bits, libFLAC adds
    // subframe->wasted_bits to each of its estimates): inside a candidate it is the same for every subframe type, between the
    // candidates of a block it is not (a DC signal: left odd, right a multiple of four) and decides the channel assignment.
    best = best > 0xFFFFFFFFu - wst ? 0xFFFFFFFFu : best + wst;
    FgPipeDec *dec = B.dec + (size_t)bi * NC + C;
    {
        // the 20 header words of the record in one coalesced store (lane j = word j; everything here is wave-uniform)
        uint32_t wv_ = 0;
        const uint32_t hw[8] = {best, d_type, d_order, d_prec, (uint32_t)d_shift, d_porder, d_method, wraw};
#pragma unroll
        for (int j = 0; j < 8; j++) wv_ = lane == j ? hw[j] : wv_;
#pragma unroll
        for (int j = 0; j < 12; j++) wv_ = lane == 8 + j ? (j < MAXO ? (uint32_t)bestq[j < MAXO ? j : 0] : 0u) : wv_;
        if (lane < 20) ((uint32_t *)dec)[lane] = wv_;
    }
    if (lane == 0) {
        FgBlockResult *r = &results[d.out_slot];
        r->best_bits[C] = best;
        if (C == 0) {
            r->bytes = 0; r->ca = 0; r->err = range_err; r->reserved = 4;
            if (range_err && B.guard) atomicOr(&B.guard[2], (unsigned long long)range_err);
#pragma unroll
            for (int c = NC; c < 4; c++) r->best_bits[c] = 0;
            for (int w = 0; w < 4; w++) B.chunk_bits[(size_t)d.out_slot * 4 + w] = 0;     // the packing waves fill in theirs
        }
    }
    if (d_type >= 2 && d_kvalid) dec->k[d_kpart] = (uint8_t)d_k;
    if (mydbg) {
        if (lane == 0) {
            FgDebugCand *dc = &mydbg->cand[C];
            dc->wasted = wst; dc->sbps = sb; dc->type = d_type; dc->order = d_type >= 2 ? d_order : 0;
            dc->precision = d_type == 3 ? d_prec : 0; dc->shift = d_type == 3 ? d_shift : 0;
            dc->bits = best; dc->porder = d_type >= 2 ? d_porder : 0; dc->rice_method = d_type >= 2 ? d_method : 0;
            for (uint32_t j = 0; j < FG_MAX_ORDER; j++) dc->qlp[j] = (d_type == 3 && j < d_order && j < (uint32_t)MAXO) ? bestq[j < (uint32_t)MAXO ? j : 0] : 0;
        }
        if (d_type >= 2 && d_kvalid) mydbg->cand[C].rice_params[d_kpart] = d_k;
    }
}

// ================================================================================================ K4w: candidates of true 32-bit content
// pipe_eval_cand for blocks of a 32-bit stream that the shifted 25-bit forms cannot hold (round 4): the same stages with the
// candidate's samples as doubles (pipe_cdbl: up to 33 bits, exact), regular lane geometry only.
//   * fixed predictors: differences and |.| sums in fp64 (differences below 2^37, sums below 2^49: exact).  From 28 bits per
//     sample on libFLAC takes its _limit_residual variants (oracle/flac_oracle.c fixed_best_predictor has the rules as the
//     reference binary shows them): the sums include the warm-up positions, an order with a residual beyond 31 bits is out, the
//     order is chosen from k = 4 down with <=, and every order's estimate comes from the order-0 total.
//   * LPC: the fp64 FIR of the 24-bit path -- products below 2^46, sums of twelve below 2^50 --, |residual| < 2^31 or the
//     candidate is dropped (libFLAC's _limit_residual FIR).
// Round 6: the ragged lane geometry too (RAG: tail blocks, odd block sizes -- PipeGeo; 23 % of the fuzz corpus's 32-bit blocks went
// to the generic kernel for want of it).  As in pipe_eval_cand<..., RAG>: every lane walks the `base` samples all working lanes
// have, the first `extra` lanes of a group one more (ln.len), idle lanes walk whatever row 0 holds and count for nothing; the
// samples in front of a lane lie at the end of ONE other lane's row (prv*).  Where the block's length past the warm-up is no multiple
// of four the reference binary's AVX2 routines sum the fixed predictors' errors their own way: reproduced below (the first version
// of these forms summed exactly, and 33 of 2949 fuzz cases on the emulator came out with another predictor order).
template <bool MS, int NCH, int MAXO, bool RAG>
FGI void pipe_eval_cand_w32(const uint32_t C, const FgBlockDesc &d, uint32_t bi, const FgEncParams &P, const FgPipeBufs &B, FgBlockResult *results,
                            FgDebugRec *mydbg, const LDS int32_t *sL, const LDS int32_t *sR, int lane, uint32_t range_err, const PipeGeo &geo)
{
    constexpr int NC = MS ? 4 : NCH;
    constexpr uint32_t PADE = PipeTypes<true>::PADE;
    const uint32_t n = d.n, seg = RAG ? geo.base : n >> 6, rstr = RAG ? geo.rstr : seg + PADE;
    (void)PADE;
    const PipeLane ln = pipe_lane<RAG>(geo, (uint32_t)lane, seg);
    const LDS int32_t *rowL = sL + (uint32_t)lane * rstr, *rowR = sR + (uint32_t)lane * rstr;
    const LDS int32_t *prvL = sL + ln.prow * rstr + ln.plen, *prvR = sR + ln.prow * rstr + ln.plen;      // one past the samples in front
    const uint32_t wraw = rfl(B.wasted[bi * NC + C]);
    const uint32_t wst = wraw & 0xFFu;
    const uint32_t nominal = P.bps + ((MS && C == 3) ? 1u : 0u);
    const uint32_t sb = nominal - wst;
    const double wscale = (wraw & 0x100u) ? 1.0 : __hiloint2double((int)((1023u - wst) << 20), 0);      // 2^-wasted
    auto samp = [&](int s) __attribute__((always_inline)) -> double { return pipe_cdbl(rowL[s], (NCH == 2) ? (int32_t)rowR[s] : 0, C, MS, wscale); };
    // the k-th sample in front of this lane's first one (k >= 1)
    auto hsamp = [&](int k) __attribute__((always_inline)) -> double {
        if (!RAG) return samp((int)seg - k - (int)rstr);
        return pipe_cdbl(prvL[-k], (NCH == 2) ? (int32_t)prvR[-k] : 0, C, MS, wscale);
    };
    auto bcast0 = [&](double v) __attribute__((always_inline)) -> double {
        return __hiloint2double((int)rl((uint32_t)__double2hiint(v), 0), (int)rl((uint32_t)__double2loint(v), 0));
    };

    uint32_t pmax0 = 0;
    { uint32_t b = n; while (!(b & 1)) { pmax0++; b >>= 1; } if (pmax0 > 15) pmax0 = 15; }
    if (P.max_po < pmax0) pmax0 = P.max_po;
    const uint32_t pmin0 = P.min_po < pmax0 ? P.min_po : pmax0;

    // ================================================================ fixed-predictor error sums
    u64 tot[5];
    u64 fsum = 0;
    uint32_t guess;
    float rbg, rb1;
    {
        double facc[5], fwarm[5], fmx[5];
        uint32_t fov[5] = {0, 0, 0, 0, 0};       // (RAG: how many of the lane's errors lie beyond 31 bits -- the AVX2 correction below needs the count)
#pragma unroll
        for (int kk = 0; kk < 5; kk++) { facc[kk] = 0.0; fwarm[kk] = 0.0; fmx[kk] = 0.0; }
        double P0 = 0.0, P1 = 0.0, P2 = 0.0, P3 = 0.0;
#pragma unroll
        for (int s = -4; s < 4; s++) {
            double v = 0.0;
            if (s >= 0) v = samp(s);
            else if (lane > 0) v = hsamp(-s);
            const double e1 = v - P0, e2 = e1 - P1, e3 = e2 - P2, e4 = e3 - P3;
            if (s >= 0) {
                const double ab[5] = {__builtin_fabs(v), __builtin_fabs(e1), __builtin_fabs(e2), __builtin_fabs(e3), __builtin_fabs(e4)};
#pragma unroll
                for (int kk = 0; kk < 5; kk++) {
                    if (lane > 0) { facc[kk] += ab[kk]; fmx[kk] = __builtin_fmax(fmx[kk], ab[kk]); if (RAG) fov[kk] += ab[kk] > 2147483647.0 ? 1u : 0u; }
                    else if (s >= kk) { fwarm[kk] += ab[kk]; fmx[kk] = __builtin_fmax(fmx[kk], ab[kk]); if (RAG) fov[kk] += ab[kk] > 2147483647.0 ? 1u : 0u; }
                }
            }
            P0 = v; P1 = e1; P2 = e2; P3 = e3;
        }
#pragma unroll 1
        for (int s = 4; s < (int)seg; s++) {
            const double v = samp(s);
            const double e1 = v - P0, e2 = e1 - P1, e3 = e2 - P2, e4 = e3 - P3;
            const double ab[5] = {__builtin_fabs(v), __builtin_fabs(e1), __builtin_fabs(e2), __builtin_fabs(e3), __builtin_fabs(e4)};
#pragma unroll
            for (int kk = 0; kk < 5; kk++) { facc[kk] += ab[kk]; fmx[kk] = __builtin_fmax(fmx[kk], ab[kk]); if (RAG) fov[kk] += ab[kk] > 2147483647.0 ? 1u : 0u; }
            P0 = v; P1 = e1; P2 = e2; P3 = e3;
        }
        if (RAG) {
            // the one sample more that the first lanes of a group have; idle lanes have walked over whatever row 0 holds
            const double v = samp((int)seg);
            const double e1 = v - P0, e2 = e1 - P1, e3 = e2 - P2, e4 = e3 - P3;
            const double ab[5] = {__builtin_fabs(v), __builtin_fabs(e1), __builtin_fabs(e2), __builtin_fabs(e3), __builtin_fabs(e4)};
#pragma unroll
            for (int kk = 0; kk < 5; kk++) {
                if (ln.len > seg) { facc[kk] += ab[kk]; fmx[kk] = __builtin_fmax(fmx[kk], ab[kk]); fov[kk] += ab[kk] > 2147483647.0 ? 1u : 0u; }
                if (!ln.act) { facc[kk] = 0.0; fwarm[kk] = 0.0; fmx[kk] = 0.0; fov[kk] = 0; }
            }
        }
        bool over[5];
        u64 warm0[5];
#pragma unroll
        for (int kk = 0; kk < 5; kk++) {
            tot[kk] = wave_sum64((u64)facc[kk]);                   // samples 4 .. n - 1
            warm0[kk] = (u64)bcast0(fwarm[kk]);                    // lane 0: samples kk .. 3
            over[kk] = __any(fmx[kk] > 2147483647.0);
        }
        if constexpr (RAG) {
            // The reference binary's AVX2 routines (oracle/flac_oracle.c avx2_lane_sums; pipe_eval_cand has the same correction for
            // its integer forms): four lanes of q = len / 4 errors that START at (j len) / 4 but take their history from j q.  With
            // len % 4 != 0 lanes 2 and 3 start late against their history: a sample or two between the lanes is not counted and
            // the first four errors of a shifted lane are not the true ones.  Below 28 bits (_wide) the len % 4 samples at the end are
            // not counted either; from 28 to 32 bits (_limit_residual) a scalar loop adds them -- and the one or two of them the
            // last lane has reached already count twice.  The 33-bit side channel takes plain C: exact.  The sums above are the
            // exact ones; every lane computes the same correction from a dozen samples.
            const uint32_t flen = n - 4, qq = flen >> 2, rr = flen & 3;
            const bool wide_form = sb < 28 && sb + ilog2_32(flen * 17) >= 32, lim_form = sb >= 28 && sb <= 32;
            if (rr != 0 && (wide_form || lim_form)) {
                const uint32_t mS = 0xFFFFFFFFu / geo.S + 1, mB1 = 0xFFFFFFFFu / (geo.base + 1) + 1, mB = 0xFFFFFFFFu / geo.base + 1;
                auto xat = [&](uint32_t g) -> i64 {          // candidate value of sample g of the block
                    const uint32_t ad = pipe_rag_addr(geo, g, mS, mB1, mB);
                    return (i64)pipe_cdbl(sL[ad], (NCH == 2) ? (int32_t)sR[ad] : 0, C, MS, wscale);
                };
                auto aabs = [](i64 v) -> u64 { return (u64)(v < 0 ? -v : v); };
                auto exact = [&](uint32_t i, u64 (&e)[5]) {   // errors of orders 0..4 at d[i] (= sample i + 4 of the block)
                    const i64 a = xat(i + 4), b = xat(i + 3), c = xat(i + 2), dd = xat(i + 1), ee = xat(i);
                    e[0] = aabs(a); e[1] = aabs(a - b); e[2] = aabs(a - 2 * b + c); e[3] = aabs(a - 3 * b + 3 * c - dd);
                    e[4] = aabs(a - 4 * b + 6 * c - 4 * dd + ee);
                };
                i64 adj[5] = {0, 0, 0, 0, 0};
                // (over[]: an order is out when one of the errors the routine COUNTS lies beyond 31 bits -- the lanes counted such errors
                // among all samples, fov; the ones it skips or replaces go out of the count, the ones it makes come in)
                int32_t nov[5];
#pragma unroll
                for (int k = 0; k < 5; k++) nov[k] = (int32_t)wave_sum(fov[k]);
                const uint32_t st2 = flen >> 1, st3 = (3 * flen) >> 2;
                auto skip = [&](uint32_t lo, uint32_t hi) {
                    for (uint32_t i = lo; i < hi; i++) { u64 e[5]; exact(i, e); for (int k = 0; k < 5; k++) { adj[k] -= (i64)e[k]; if (e[k] > 0x7FFFFFFFull) nov[k]--; } }
                };
                skip(2 * qq, st2); skip(st2 + qq, st3);
                if (wide_form) skip(st3 + qq, flen);
                else for (uint32_t i = 4 * qq; i < st3 + qq; i++) { u64 e[5]; exact(i, e); for (int k = 0; k < 5; k++) adj[k] += (i64)e[k]; }
                auto shifted = [&](uint32_t j, uint32_t st) {
                    const uint32_t hb = j * qq;
                    if (st == hb) return;
                    const i64 h1 = xat(hb + 3), h2 = xat(hb + 2), h3 = xat(hb + 1), h4 = xat(hb);
                    i64 p0 = h1, p1 = h1 - h2, p2 = p1 - (h2 - h3), p3 = p2 - (h2 - 2 * h3 + h4);
                    for (uint32_t i = 0; i < 4 && i < qq; i++) {
                        const i64 e0 = xat(st + i + 4), e1 = e0 - p0, e2 = e1 - p1, e3 = e2 - p2, e4 = e3 - p3;
                        const i64 ev[5] = {e0, e1, e2, e3, e4};
                        u64 ex[5];
                        exact(st + i, ex);
                        for (int k = 0; k < 5; k++) {
                            adj[k] += (i64)aabs(ev[k]) - (i64)ex[k];
                            if (ex[k] > 0x7FFFFFFFull) nov[k]--;
                            if (aabs(ev[k]) > 0x7FFFFFFFull) nov[k]++;
                        }
                        p3 = e3; p2 = e2; p1 = e1; p0 = e0;
                    }
                };
                shifted(2, st2); shifted(3, st3);
#pragma unroll
                for (int kk = 0; kk < 5; kk++) { tot[kk] = (u64)((i64)tot[kk] + adj[kk]); over[kk] = nov[kk] > 0; }
            }
        }
        const double len = (double)(n - 4);
        if (sb < 28) {
            const u64 m34 = tot[3] < tot[4] ? tot[3] : tot[4];
            const u64 m234 = tot[2] < m34 ? tot[2] : m34;
            const u64 m1234 = tot[1] < m234 ? tot[1] : m234;
            if (tot[0] <= m1234) guess = 0;
            else if (tot[1] <= m234) guess = 1;
            else if (tot[2] <= m34) guess = 2;
            else if (tot[3] <= tot[4]) guess = 3;
            else guess = 4;
            const u64 tg = guess == 0 ? tot[0] : guess == 1 ? tot[1] : guess == 2 ? tot[2] : guess == 3 ? tot[3] : tot[4];
            rbg = (float)((tg > 0) ? log(FG_LN2 * (double)tg / len) / FG_LN2 : 0.0);
            rb1 = tot[1] == 0 ? 0.0f : 1.0f;
        }
        else {
            // the _limit_residual rules (see the comment above): totals with the warm-up positions, orders from 4 down
#pragma unroll
            for (int kk = 0; kk < 5; kk++) tot[kk] += warm0[kk];
            const float r0 = (float)((tot[0] > 0) ? log(FG_LN2 * (double)tot[0] / len) / FG_LN2 : 0.0);
            u64 smallest = ~(u64)0;
            guess = 0;
            rbg = 34.0f; rb1 = 34.0f;
            float rbk[5];
#pragma unroll
            for (int kk = 4; kk >= 0; kk--) {
                if (!over[kk] && tot[kk] <= smallest) { guess = (uint32_t)kk; smallest = tot[kk]; rbk[kk] = r0; }
                else rbk[kk] = 34.0f;
            }
            rbg = guess == 0 ? rbk[0] : guess == 1 ? rbk[1] : guess == 2 ? rbk[2] : guess == 3 ? rbk[3] : rbk[4];
            rb1 = rbk[1];
        }
        const double a = guess == 0 ? facc[0] : guess == 1 ? facc[1] : guess == 2 ? facc[2] : guess == 3 ? facc[3] : facc[4];
        const double w = guess == 0 ? fwarm[0] : guess == 1 ? fwarm[1] : guess == 2 ? fwarm[2] : guess == 3 ? fwarm[3] : fwarm[4];
        fsum = (u64)(lane == 0 ? a + w : a);
    }

    // ---- baseline: verbatim / constant
    uint32_t best;
    uint32_t d_type = 1, d_order = 0, d_prec = 0, d_porder = 0, d_method = 0, d_k = 0, d_kpart = 0;
    bool d_kvalid = false;
    int d_shift = 0;
    int32_t bestq[MAXO];
#pragma unroll
    for (int j = 0; j < MAXO; j++) bestq[j] = 0;
    bool do_fixed = false, do_lpc = false;
    {
        const u64 vb = (u64)8 + (u64)n * sb;
        best = vb < 0xFFFFFFFFull ? (uint32_t)vb : 0xFFFFFFFFu;
        bool constant = false;
        if (rb1 == 0.0f) {
            const double x0 = pipe_cdbl(sL[0], (NCH == 2) ? (int32_t)sR[0] : 0, C, MS, wscale);
            uint32_t ne = 0;
#pragma unroll 1
            for (uint32_t s = 0; s < seg + (RAG ? 1u : 0u); s++) ne |= ((!RAG || s < ln.len) && samp((int)s) != x0);
            constant = !__any(ne != 0);
        }
        if (mydbg && lane == 0) {
            for (int kk = 0; kk < 5; kk++) mydbg->cand[C].fixed_tot[kk] = tot[kk];
            mydbg->cand[C].fixed_guess = guess;
        }
        if (P.limit_min_bitrate && C >= (uint32_t)(NCH - 1) && d.forced_ca != 3) {
            bool forbid = true;
            if (NCH == 2) {
                const int32_t l0 = sL[0];
                uint32_t ne = 0;
#pragma unroll 1
                for (uint32_t s = 0; s < seg + (RAG ? 1u : 0u); s++) ne |= ((!RAG || s < ln.len) && (int32_t)rowL[s] != l0);
                forbid = !__any(ne != 0);
                // ("chose CONSTANT", not "is constant": at 28 bits and more only an all-zero left channel does -- see pipe_eval_cand)
                const uint32_t wl = rfl(B.wasted[(size_t)bi * NC]);
                if (forbid && !(wl & 0x100u) && P.bps - (wl & 0xFFu) >= 28) forbid = false;
            }
            if (forbid) constant = false;
        }
        if (constant) {
            const uint32_t cb = 8 + sb;
            if (cb < best) { best = cb; d_type = 0; }
        }
        else {
            if (!(rbg >= (float)sb)) do_fixed = true;
            if (P.max_lpc_order > 0) do_lpc = true;
        }
    }
    const uint32_t nv = do_lpc ? rfl(B.nv[bi]) : 0;
    const uint32_t limit = P.rice_limit;

#pragma unroll 1
    for (uint32_t pass = 0; pass < 1 + nv; pass++) {
        uint32_t order, prec = 0;
        int shift = 0;
        int32_t q[MAXO];
#pragma unroll
        for (int j = 0; j < MAXO; j++) q[j] = 0;
        u64 psum;
        uint32_t ovf = 0;
        const int kind = pass == 0 ? 0 : 1;
        if (pass == 0) {
            if (!do_fixed) continue;
            order = guess;
            psum = fsum;
        }
        else {
            const uint32_t v = pass - 1;
            const size_t ridx = ((size_t)bi * NC + C) * P.nvec + v;
            const uint32_t r = rfl(B.lres[ridx]);
            order = r & 0xFF; prec = (r >> 8) & 0xFF; shift = (int)(int8_t)((r >> 16) & 0xFF);
            if (mydbg && lane == 0) mydbg->cand[C].lpc_guess[v] = ((r >> 25) & 1) ? (r & 0xFF) : 0;
            if (!((r >> 24) & 1)) continue;
            if (order == 0) order = 1;
            const int32_t qall = (lane < MAXO) ? B.qres[ridx * MAXO + lane] : 0;
            double hd[MAXO], qd[MAXO];
            const double scl = __hiloint2double((int)((1023u - (uint32_t)shift) << 20), 0);      // 2^-shift
            const double qdl = (double)qall * scl;
#pragma unroll
            for (int j = 0; j < MAXO; j++) {
                q[j] = (int32_t)rl((uint32_t)qall, j);
                qd[j] = __hiloint2double((int)rl((uint32_t)__double2hiint(qdl), j), (int)rl((uint32_t)__double2loint(qdl), j));
            }
            double psumd = 0.0, pmaxd = 0.0;
            // (without the running maximum first, as in pipe_eval_cand)
            auto walk = [&](auto TRACK) __attribute__((always_inline)) {
            psumd = 0.0; pmaxd = 0.0;
#pragma unroll
            for (int j = 0; j < MAXO; j++) hd[(MAXO - 1 - j) % MAXO] = lane > 0 ? hsamp(1 + j) : 0.0;
            auto step = [&](int u, uint32_t s, bool guard) __attribute__((always_inline)) {
                const double xd = samp((int)s);
                const bool real = !guard || lane > 0 || s >= order;
                const double rr = __builtin_floor(pfir_f64n<MAXO>(qd, hd, u, -xd));          // (minus the residual)
                if (real) { psumd += __builtin_fabs(rr); if (decltype(TRACK)::value) pmaxd = __builtin_fmax(pmaxd, __builtin_fabs(rr)); }
                hd[u] = xd;
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
            if (!RAG) {
#pragma unroll
                for (int u = 0; u < MAXO; u++) if (s0 + u < seg) step(u, s0 + u, s0 == 0);
            }
            else {
                // (the rest of the lane: the common `seg` samples and the one more of the first lanes of a group; at most MAXO + 1 of them --
                // sample s0 + u sits in history slot u mod MAXO)
#pragma unroll
                for (int u = 0; u <= MAXO; u++) if (s0 + u < ln.len) step(u % MAXO, s0 + u, s0 == 0);
                if (!ln.act) { psumd = 0.0; pmaxd = 0.0; }
            }
            };
            walk(std::false_type());
            if (__any(!(psumd < 2147483648.0))) walk(std::true_type());
            // (a residual must fit int32: libFLAC's test is r <= INT32_MIN || r > INT32_MAX, i.e. |r| < 2^31 passes)
            if (!(pmaxd < 2147483648.0)) ovf = 1;
            psum = ovf ? 0ull : (u64)psumd;
        }
        const bool dead = __any(ovf != 0);
        uint32_t best_bits, bpo, kb, kpart;
        bool kvalid;
        // (a partition must be longer than the predictor order: short blocks of the ragged geometry meet that limit, see pipe_eval_cand)
        uint32_t pmax_p = pmax0;
        if (RAG) while (pmax_p > 0 && (n >> pmax_p) <= order) pmax_p--;
        const uint32_t pmin_p = pmin0 < pmax_p ? pmin0 : pmax_p;
        if (!pipe_rice_search_tree<true>(psum, lane, n, order, sb, pmin_p, pmax_p, limit, best_bits, bpo, kb, kpart, kvalid)) {
            pipe_rice_search<true>(psum, dead, lane, n, order, sb, pmin_p, pmax_p, limit, best_bits, bpo, kb);
            kvalid = ((uint32_t)lane & ((64u >> bpo) - 1)) == 0;
            kpart = (uint32_t)lane >> (6 - bpo);
        }
        uint32_t est = 0;
        if (!dead) {
            est = kind == 0 ? (8 + order * sb) : (8 + 4 + 5 + order * (prec + sb));
            if (best_bits < 0xFFFFFFFFu - est) est += best_bits; else est = 0xFFFFFFFFu;
            if (est > 0 && est < best) {
                best = est;
                d_type = kind == 0 ? 2 : 3; d_order = order; d_prec = prec; d_shift = shift;
                d_porder = bpo; d_k = kb; d_kpart = kpart; d_kvalid = kvalid;
                d_method = __any(kvalid && kb >= 15) ? 1 : 0;
                if (kind == 1) {
#pragma unroll
                    for (int j = 0; j < MAXO; j++) bestq[j] = q[j];
                }
            }
        }
        if (mydbg && lane == 0) {
            const uint32_t estw = (est == 0 || est > 0xFFFFFFFFu - wst) ? est : est + wst;
            if (kind == 0) mydbg->cand[C].fixed_bits = estw;
            else mydbg->cand[C].lpc_bits[pass - 1] = estw;
        }
    }

    // ---- the decision record of this candidate (as pipe_eval_cand writes it)
    best = best > 0xFFFFFFFFu - wst ? 0xFFFFFFFFu : best + wst;
    FgPipeDec *dec = B.dec + (size_t)bi * NC + C;
    {
        uint32_t wv_ = 0;
        const uint32_t hw[8] = {best, d_type, d_order, d_prec, (uint32_t)d_shift, d_porder, d_method, wraw};
#pragma unroll
        for (int j = 0; j < 8; j++) wv_ = lane == j ? hw[j] : wv_;
#pragma unroll
        for (int j = 0; j < 12; j++) wv_ = lane == 8 + j ? (j < MAXO ? (uint32_t)bestq[j < MAXO ? j : 0] : 0u) : wv_;
        if (lane < 20) ((uint32_t *)dec)[lane] = wv_;
    }
    if (lane == 0) {
        FgBlockResult *r = &results[d.out_slot];
        r->best_bits[C] = best;
        if (C == 0) {
            r->bytes = 0; r->ca = 0; r->err = range_err; r->reserved = 4;
            if (range_err && B.guard) atomicOr(&B.guard[2], (unsigned long long)range_err);
#pragma unroll
            for (int c = NC; c < 4; c++) r->best_bits[c] = 0;
            for (int w = 0; w < 4; w++) B.chunk_bits[(size_t)d.out_slot * 4 + w] = 0;
        }
    }
    if (d_type >= 2 && d_kvalid) dec->k[d_kpart] = (uint8_t)d_k;
    if (mydbg) {
        if (lane == 0) {
            FgDebugCand *dc = &mydbg->cand[C];
            dc->wasted = wst; dc->sbps = sb; dc->type = d_type; dc->order = d_type >= 2 ? d_order : 0;
            dc->precision = d_type == 3 ? d_prec : 0; dc->shift = d_type == 3 ? d_shift : 0;
            dc->bits = best; dc->porder = d_type >= 2 ? d_porder : 0; dc->rice_method = d_type >= 2 ? d_method : 0;
            for (uint32_t j = 0; j < FG_MAX_ORDER; j++) dc->qlp[j] = (d_type == 3 && j < d_order && j < (uint32_t)MAXO) ? bestq[j < (uint32_t)MAXO ? j : 0] : 0;
        }
        if (d_type >= 2 && d_kvalid) mydbg->cand[C].rice_params[d_kpart] = d_k;
    }
}

// (samples above 16 bits are staged as int32: 33 KB of LDS per stereo block let four workgroups share a CU, so those forms
// may use 128 registers -- the 64 that eight workgroups per CU allow cost them 25 spilled vector registers)
template <bool MS, int NCH, int MAXO, bool ACC64, bool RAG>
__global__ void __launch_bounds__((MS ? 4 : NCH) * 64, (ACC64 || RAG) ? 4 : 8)
fg_pipe_eval_kernel(const void *pcm, const FgBlockDesc *descs, FgEncParams P, FgPipeBufs B, FgBlockResult *results, FgDebugRec *dbg, uint32_t bi0)
{
    constexpr int NC = MS ? 4 : NCH;
    constexpr int NT = NC * 64;
    typedef typename PipeTypes<ACC64>::samp_t samp_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t bi = blockIdx.x + bi0;
    const FgBlockDesc d = descs[bi];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = rfl((uint32_t)tid >> 6);
    const uint32_t sbytes = ((pipe_rows_elems(P.sig_stride, 64) * (uint32_t)sizeof(samp_t)) + 15) & ~15u;
    LDS samp_t *sL = (LDS samp_t *)smem;
    LDS samp_t *sR = (LDS samp_t *)((LDS unsigned char *)smem + sbytes);
    LDS uint32_t *xch = (LDS uint32_t *)((LDS unsigned char *)smem + (NCH == 2 ? 2 : 1) * sbytes);
    if (tid == 0) xch[0] = 0;
    __syncthreads();
    const PipeGeo geo = RAG ? pipe_geo(d.n, 6, P.max_po, PipeTypes<ACC64>::PADE) : pipe_geo_regular(d.n >> 6, PipeTypes<ACC64>::PADE);
    bool pre_ok = true;
    const uint32_t pre = ACC64 ? pipe_preshift<NCH, NC>(P, B, bi, pre_ok) : 0u;
    const uint32_t bad = pipe_stage<NCH, ACC64, NT, RAG>(pcm, d, P, sL, NCH == 2 ? sR : sL, tid, d.n >> 6, geo, pre);
    if (bad) xch[0] = 1;            // (benign race: every writer stores the same value)
    __syncthreads();
    const uint32_t range_err = xch[0] ? FG_ERR_RANGE : 0;
    FgDebugRec *mydbg = dbg ? dbg + d.out_slot : nullptr;
    // (the ragged geometry -- tail blocks, odd block sizes -- is a kernel of its own: the regular one keeps its loops and registers)
    if constexpr (ACC64) {
        // (32-bit streams: a block whose candidates do not fit the shifted 25-bit forms is evaluated in fp64, pipe_eval_cand_w32)
        if (P.bps == 32 && !pre_ok) {
            pipe_eval_cand_w32<MS, NCH, MAXO, RAG>(NC == 1 ? 0u : wv, d, bi, P, B, results, mydbg, (const LDS int32_t *)sL, (const LDS int32_t *)sR, lane, range_err, geo);
            return;
        }
    }
    pipe_eval_cand<MS, NCH, MAXO, ACC64, RAG>(NC == 1 ? 0u : wv, d, bi, P, B, results, mydbg, sL, sR, lane, range_err, geo, pre, pre_ok);
}

// ================================================================================================ K5: pack
// Frame-bit window of one wave: a zeroed window of `fbw` words that starts at word `wbase` of the wave's chunk.  flush()
// writes the complete words out as they are (most significant bit first inside the 32-bit value; K6 turns them into bytes).
struct ChunkBits {
    LDS uint32_t *w;
    uint32_t *outw;
    uint32_t wbase, cap_words, fbw, err;
};

// (WG: the window is shared by the waves of the workgroup -- the direct packing path)
template <bool WG = false>
FGI void cb_or(const ChunkBits &b, uint32_t pos, uint32_t val, uint32_t vbits)
{
    const uint32_t rel = pos - (b.wbase << 5);
    const uint32_t word = rel >> 5, sh = rel & 31;
    const u64 x = (u64)val << ((64 - sh - vbits) & 63);
    if constexpr (WG) {
        __hip_atomic_fetch_or(&b.w[word], (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or(&b.w[word + 1], (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    else {
        __hip_atomic_fetch_or(&b.w[word], (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_or(&b.w[word + 1], (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

// write out the complete words below bit position `upto` (and, with `all`, the partial word that follows)
FGI void cb_flush(ChunkBits &b, int lane, uint32_t upto, bool all)
{
    const uint32_t wend = (upto >> 5) + ((all && (upto & 31)) ? 1u : 0u);
    if (wend <= b.wbase) return;
    const uint32_t nfull = wend - b.wbase;
    if (wend > b.cap_words) b.err |= FG_ERR_SLOT;
    wave_lds_fence();
    // rows of 64 words aligned in the chunk (256-byte aligned in memory): whole cache lines except at the two ends
    for (uint32_t row = b.wbase & ~63u; row < wend; row += 64) {
        const uint32_t wi = row + (uint32_t)lane;
        if (wi >= b.wbase && wi < wend && wi < b.cap_words) b.outw[wi] = b.w[wi - b.wbase];
    }
    const uint32_t carry = b.w[nfull];
    wave_lds_fence();
    for (uint32_t j = lane; j <= nfull + 1 && j < b.fbw + 2; j += 64) b.w[j] = 0;
    wave_lds_fence();
    if (lane == 0) b.w[0] = carry;
    b.wbase = wend;
    wave_lds_fence();
}

FGI void cb_reserve(ChunkBits &b, int lane, uint32_t bitpos, uint32_t bits)
{
    if (bitpos + bits - (b.wbase << 5) > 32u * b.fbw - 64u) cb_flush(b, lane, bitpos, false);
}


// ---- decoupled look-back over the frame sizes of a call (FgPackDirect, fg_types.h).  Words are written and read whole (64-bit
// relaxed atomics at device scope: the L2 caches of the eight XCDs are not coherent for plain accesses), so a word is either of
// this call -- then state and value belong together -- or counts as empty.
FGI u64 lb_word(uint32_t epoch, u64 state, u64 value) { return ((u64)(epoch & 0xFFFFFu) << 44) | (state << FG_LB_VBITS) | (value & ((1ull << FG_LB_VBITS) - 1)); }
FGI void lb_publish(u64 *lb, uint32_t k, u64 w) { __hip_atomic_store(&lb[k], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Sum of the sizes of frames 0 .. k - 1 (one wave, all lanes; the result is wave-uniform).  A trip to memory fetches FG_LB_NWIN
// windows of 64 frames (lane l: frames pos - l, pos - 64 - l, ...), examined nearest first; lb_fetch only issues the loads -- the
// packing kernel does so ahead of its CRC pass and looks at the words behind it, so the first trip costs nothing.  The workgroups of
// a launch are resident a thousand at a time and publish their prefix late in their lives, so the nearest prefix usually lies a few
// windows back and the later trips are waited for.  More windows a trip do NOT pay: the loads go past the L2 (device scope), and on
// the headline stream the encode launch took 0.432 / 0.434 / 0.451 / 0.449 / 0.451 ms with 1 / 2 / 4 / 8 / 16 windows a trip (same
// box, gpurun_exp w1..w8); 0.443 with one window and no fetch ahead, 0.426 with no look-back at all (a tuning build: wrong places).
// An empty word in front of the first prefix is waited for (its workgroup runs, or will: workgroups start in the order of their
// numbers); a poisoned word, or a wait of more than 5 ms (a whole call takes less), gives up.
#ifndef FG_LB_NWIN
#define FG_LB_NWIN 1
#endif
FGI void lb_fetch(const u64 *lb, int64_t pos, uint32_t epoch, int lane, u64 (&v)[FG_LB_NWIN])
{
#pragma unroll
    for (int w = 0; w < FG_LB_NWIN; w++) {
        const int64_t idx = pos - 64 * w - lane;
        v[w] = lb_word(epoch, FG_LB_PFX, 0);                       // in front of frame 0: nothing
        if (idx >= 0) v[w] = __hip_atomic_load(&lb[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// 0: done (acc has the sum); 1: sizes only, look further back (acc and pos moved); 2: an empty word in the way (nothing moved: fetch
// again); 3: poisoned
FGI int lb_examine(const u64 (&v)[FG_LB_NWIN], uint32_t epoch, int lane, u64 &acc, int64_t &pos)
{
    const u64 vmask = (1ull << FG_LB_VBITS) - 1;
    u64 mine = 0;
#pragma unroll
    for (int w = 0; w < FG_LB_NWIN; w++) {
        const uint32_t st = ((uint32_t)(v[w] >> 44) == (epoch & 0xFFFFFu)) ? ((uint32_t)(v[w] >> FG_LB_VBITS) & 3u) : 0u;
        const u64 m_inv = __ballot(st == 0), m_pfx = __ballot(st == (uint32_t)FG_LB_PFX), m_poi = __ballot(st == (uint32_t)FG_LB_POISON);
        const uint32_t p = m_pfx ? (uint32_t)__builtin_ctzll(m_pfx) : 64u;
        const u64 upto = p >= 63 ? ~0ull : ((2ull << p) - 1);
        if (m_poi & upto) return 3;
        if (m_inv & upto) return 2;
        mine += ((uint32_t)lane <= p) ? (v[w] & vmask) : 0ull;
        if (p < 64) { acc += wave_sum64(mine); return 0; }
    }
    acc += wave_sum64(mine);
    pos -= 64 * FG_LB_NWIN;
    return 1;
}
// the rest of a look-back whose first fetch is in `v`
FGI bool lb_finish(const u64 *lb, uint32_t epoch, int lane, u64 (&v)[FG_LB_NWIN], int64_t pos, u64 &excl, bool spin)
{
    u64 acc = 0;
    const u64 t0 = wall_clock64();
    for (;;) {
        const int r = lb_examine(v, epoch, lane, acc, pos);
        if (r == 0 || pos < 0) break;
        if (r == 3) return false;
        if (r == 2) {
            if (!spin || wall_clock64() - t0 > 500000ull) return false;
            __builtin_amdgcn_s_sleep(4);
        }
        lb_fetch(lb, pos, epoch, lane, v);
    }
    excl = acc;
    return true;
}
FGI bool lb_lookback(const u64 *lb, uint32_t k, uint32_t epoch, int lane, u64 &excl, bool spin)
{
    u64 v[FG_LB_NWIN];
    excl = 0;
    if (k == 0) return true;
    lb_fetch(lb, (int64_t)k - 1, epoch, lane, v);
    return lb_finish(lb, epoch, lane, v, (int64_t)k - 1, excl, spin);
}

// Blocks that keep the chunk form beside a direct launch (short blocks, the ragged geometry, frames of the generic kernel) publish
// their sizes here, in front of the direct packing kernels that will look them up.  Thread = block of descs[first, first + count).
__global__ void __launch_bounds__(256)
fg_pipe_publish_kernel(const FgBlockDesc *descs, uint32_t first, uint32_t count, const FgBlockResult *results, const uint32_t *chunk_bits,
                       FgPackDirect D)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t k = descs[first + i].out_slot;
    const FgBlockResult r = results[k];
    u64 w;
    if (r.err & FG_ERR_REDO) w = lb_word(D.epoch, FG_LB_POISON, 0);
    else {
        uint32_t nb = r.bytes;
        if (r.reserved == 4) {
            const uint4 cb = *(const uint4 *)&chunk_bits[(size_t)k * 4];
            nb = ((cb.x + cb.y + cb.z + cb.w + 7) >> 3) + 2;
        }
        w = lb_word(D.epoch, k == 0 ? FG_LB_PFX : FG_LB_AGG, nb);
    }
    lb_publish(D.lb, k, w);
}

// KEEP (round 4; <= 16-bit input, two waves per subframe, blocks of 4096: 32 samples a lane): the first walk over a lane's samples --
// the one that measures the codes -- keeps the zig-zagged residuals, two to a register, and the second walk writes the codes from
// them: the FIR, the candidate arithmetic and the LDS reads of the samples happen once instead of twice (51 -> 38 instructions a
// sample).  Sixteen more registers: four workgroups per CU instead of five.  A residual beyond 16 bits, a verbatim subframe or a
// frame-bit window too small for all 64 lanes at once takes the two-walk form as before.
// DIRECT (round 5): the workgroup's waves write into ONE frame buffer in LDS at their final bit positions (the four chunk lengths
// are exchanged behind the measuring walk), take the CRC-16 of the frame there and store the bytes at the frame's final place in
// the output stream, which a decoupled look-back over the frame sizes supplies (FgPackDirect).  Two waves per subframe only.
// ALIAS (with DIRECT and KEEP; blocks of 4096 samples): the frame buffer lies OVER the staged samples.  The measuring walk keeps the
// residuals in registers, so behind it nobody needs the samples any more -- 22 KB of LDS a workgroup instead of 39.5, and the
// registers then allow five workgroups a CU instead of four.  A wave that has to walk its samples a second time (a verbatim
// subframe, a residual beyond 16 bits, a short block) reads them from memory there: rare, and then slow.
// FUSED (with DIRECT, KEEP and ALIAS): the evaluation of the block's candidates runs in the same workgroup in front of the packing --
// wave = candidate, as in fg_pipe_eval_kernel --, on ONE staging of the samples in the evaluation's rows (64 rows of n / 64); the
// packer's lanes then own half rows.  The block's PCM is read once instead of twice and one kernel boundary goes (VERDICT round 4,
// item 1).  Five workgroups a CU for both parts, where the evaluation alone runs eight.
template <bool MS, int NCH, int MAXO, bool ACC64, int WS, bool RAG, bool KEEP = false, bool DIRECT = false, bool ALIAS = false, bool FUSED = false>
#ifndef FGX_ALIAS_WAVES
#define FGX_ALIAS_WAVES 5
#endif
__global__ void __launch_bounds__(NCH * WS * 64, ACC64 ? 3 : (ALIAS ? FGX_ALIAS_WAVES : ((RAG || KEEP || DIRECT) ? 4 : 5)))
fg_pipe_pack_kernel(const void *pcm, const FgBlockDesc *descs, FgEncParams P, FgPipeBufs B, uint8_t *slots, FgBlockResult *results,
                    uint32_t chunk_cap_words, uint32_t fbw_words, uint32_t bi0, FgPackDirect D)
{
    constexpr int NC = MS ? 4 : NCH;
    constexpr int NW = NCH * WS;            // waves launched per block
    constexpr int NT = NW * 64;
    typedef typename PipeTypes<ACC64>::samp_t samp_t;
    constexpr uint32_t PADE = PipeTypes<ACC64>::PADE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t bi = blockIdx.x + bi0;
    const FgBlockDesc d = descs[bi];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = rfl((uint32_t)tid >> 6);
    const uint32_t n = d.n;
    // waves per subframe of THIS block: WS when a lane then still walks >= 32 samples, else one (short blocks -- normally the
    // tail blocks of the streams -- ride in the same launch; their spare waves leave after the staging)
    const uint32_t ws = (WS == 2 && n % 128 == 0 && n / 128 >= 32) ? 2u : 1u;
    const uint32_t LPS = 64 * ws;           // lanes (segments) per subframe
    // (DIRECT: the spare waves of a block packed by one wave per subframe stay for the barriers and the CRC pass; they code nothing)
    const bool spare = DIRECT && wv >= (uint32_t)NCH * ws;
    const uint32_t si = spare ? 0u : wv / ws, hf = spare ? 1u : wv % ws;
    // lane geometry (see PipeGeo): ragged blocks are packed by one wave per subframe
    const PipeGeo geo = RAG ? pipe_geo(n, 6, P.max_po, PADE) : pipe_geo_regular(n / LPS, PADE);
    constexpr bool rag = RAG;
    // (FUSED: the rows are the evaluation's -- 64 of n / 64 samples, ws lanes of the packer to a row)
    const PipeGeo geoE = pipe_geo_regular(n >> 6, PADE);
    const uint32_t seg = rag ? geo.base : n / LPS, rstr = FUSED ? geoE.rstr : geo.rstr;
    const uint32_t sbytes = ((pipe_rows_elems(P.sig_stride, FUSED ? 64 : 64 * WS) * (uint32_t)sizeof(samp_t)) + 15) & ~15u;
    LDS samp_t *sL = (LDS samp_t *)smem;
    LDS samp_t *sR = (LDS samp_t *)((LDS unsigned char *)smem + sbytes);
    // (DIRECT: ONE window of fbw_words = the whole frame, shared by the waves; behind it the waves' scratch words, the words the waves
    // exchange and the two look-up tables of the CRC pass)
    LDS uint32_t *const area = (LDS uint32_t *)((LDS unsigned char *)smem + (NCH == 2 ? 2 : 1) * sbytes);
    // (ALIAS: the frame buffer starts where the staged samples do; the rest lies behind the longer of the two)
    const uint32_t stg_words = (uint32_t)(((NCH == 2 ? 2 : 1) * sbytes) >> 2);
    LDS uint32_t *const tailbase = ALIAS ? (LDS uint32_t *)smem + (stg_words > fbw_words + 2 ? stg_words : fbw_words + 2) : area + (fbw_words + 2);
    LDS uint32_t *fbw = ALIAS ? (LDS uint32_t *)smem : (DIRECT ? area : area + wv * (fbw_words + 2 + 64));
    LDS uint32_t *misc = DIRECT ? tailbase + wv * 64 : fbw + fbw_words + 2;     // 64 words per wave: header bytes, then the packer's scratch words
    LDS uint32_t *const xch = tailbase + NW * 64;                             // (DIRECT) 16 words
    LDS uint16_t *const ctab = (LDS uint16_t *)(xch + 16);                     // (DIRECT) 1536 entries
    bool pre_ok = true;
    const uint32_t pre = ACC64 ? pipe_preshift<NCH, NC>(P, B, bi, pre_ok) : 0u;
    uint32_t stage_bad = 0;
    if constexpr (FUSED) stage_bad = pipe_stage<NCH, ACC64, NT, false>(pcm, d, P, sL, NCH == 2 ? sR : sL, tid, n >> 6, geoE, pre);
    else (void)pipe_stage<NCH, ACC64, NT, RAG>(pcm, d, P, sL, NCH == 2 ? sR : sL, tid, seg, geo, pre);
    if constexpr (DIRECT) {
        if constexpr (!ALIAS) { for (uint32_t j = tid; j < fbw_words + 2; j += NT) fbw[j] = 0; }
        for (uint32_t j = tid; j < 1536; j += NT) ctab[j] = D.crcx[j];
        if (tid < 16) xch[tid] = 0;
    }
    else { for (uint32_t j = lane; j < fbw_words + 2; j += 64) fbw[j] = 0; }
    __syncthreads();
    if (!DIRECT && wv >= (uint32_t)NCH * ws) return;           // (no barrier after this point)
    if constexpr (FUSED) {
        // ---- the evaluation (fg_pipe_eval_kernel's body): wave = candidate; its decisions go to B.dec as ever and are read back below
        if (stage_bad) xch[15] = 1;         // (benign race: every writer stores the same value)
        __syncthreads();
        const uint32_t range_err = xch[15] ? FG_ERR_RANGE : 0;
        if (wv < (uint32_t)NC) pipe_eval_cand<MS, NCH, MAXO, false, false>(NC == 1 ? 0u : wv, d, bi, P, B, results, nullptr, sL, NCH == 2 ? sR : sL, lane, range_err, geoE, 0, true);
        __threadfence_block();
        __syncthreads();
    }

    // ---- channel assignment from the four candidate totals (every wave computes it; wave-uniform)
    uint32_t ca = 0, c = si;
    // (32-bit streams: bit 31 of a candidate's size = the evaluation could not take this block -- pipe_preshift --: generic kernel)
    bool unsure = false;
    if (ACC64 && P.bps == 32) {
#pragma unroll
        for (int k = 0; k < NC; k++) unsure = unsure || (rfl(B.dec[(size_t)bi * NC + k].bits) >> 31) != 0;
    }
    if (unsure) {
        // (every wave of the workgroup sees the same flags and leaves here.  Since round 6 the evaluation flags nothing -- a block the
        // shifted forms cannot hold takes pipe_eval_cand_w32 --; what stays is the hand-over itself, and with DIRECT the word the frames
        // behind this one would otherwise wait for)
        if (lane == 0) {
            B.chunk_bits[(size_t)d.out_slot * 4 + wv] = 0;
            atomicOr(&results[d.out_slot].err, FG_ERR_REDO);
            if (B.guard) atomicOr(&B.guard[2], (unsigned long long)(FG_ERR_REDO | (DIRECT ? FG_ERR_CHAIN : 0u)));
            if (DIRECT && wv == 0) lb_publish(D.lb, d.out_slot, lb_word(D.epoch, FG_LB_POISON, 0));
        }
        return;
    }
    if (MS) {
        if (d.forced_ca != 0xFF) ca = d.forced_ca & 0x7F;       // (bit 7: a loose mid-side DECISION frame, see FgBlockDesc)
        else {
            const uint32_t b0 = rfl(B.dec[(size_t)bi * NC + 0].bits), b1 = rfl(B.dec[(size_t)bi * NC + (NC > 1 ? 1 : 0)].bits);
            const uint32_t b2 = rfl(B.dec[(size_t)bi * NC + (NC > 2 ? 2 : 0)].bits), b3 = rfl(B.dec[(size_t)bi * NC + (NC > 3 ? 3 : 0)].bits);
            const uint32_t b01 = b0 + b1, b03 = b0 + b3, b13 = b1 + b3, b23 = b2 + b3;
            uint32_t mn = b01;
            if (b03 < mn) { mn = b03; ca = 1; }
            if (b13 < mn) { mn = b13; ca = 2; }
            if (b23 < mn) { mn = b23; ca = 3; }
        }
        const uint32_t sub0 = ca == 2 ? 3 : (ca == 3 ? 2 : 0), sub1 = ca == 0 ? 1 : (ca == 2 ? 1 : 3);
        c = si == 0 ? sub0 : sub1;
    }
    const FgPipeDec *dec = B.dec + (size_t)bi * NC + c;
    const uint32_t type = spare ? 0u : rfl(dec->type), order = rfl(dec->order), prec = rfl(dec->prec), po = rfl(dec->porder), method = rfl(dec->method);
    const uint32_t wraw = rfl(dec->wasted), wst = wraw & 0xFFu;
    int shift = (int)rfl((uint32_t)dec->shift);
    const uint32_t sb = P.bps + ((MS && c == 3) ? 1u : 0u) - wst;
    const uint32_t mask = sb < 32 ? ((1u << sb) - 1) : 0xFFFFFFFFu;
    const uint32_t Lg = hf * 64 + (uint32_t)lane;         // this lane's segment of the subframe
    // (FUSED: ws lanes share one of the evaluation's rows; hoff: where the sample in front of the lane's first one lies, from its row)
    const LDS samp_t *rowL = FUSED ? sL + (Lg / ws) * rstr + (Lg % ws) * seg : sL + Lg * rstr;
    const LDS samp_t *rowR = FUSED ? sR + (Lg / ws) * rstr + (Lg % ws) * seg : sR + Lg * rstr;
    const int hoff = (FUSED && (Lg % ws) != 0) ? -1 : (int)(FUSED ? ws * seg : seg) - 1 - (int)rstr;
    const PipeLane ln = pipe_lane<RAG>(geo, Lg, seg);
    const LDS samp_t *prvL = sL + ln.prow * rstr + ln.plen, *prvR = sR + ln.prow * rstr + ln.plen;      // one past the samples in front

    ChunkBits fb;
    fb.w = fbw; fb.fbw = fbw_words; fb.cap_words = chunk_cap_words; fb.wbase = 0; fb.err = 0;
    const uint32_t chunk = si * 2 + hf;            // chunk slots in bit order: (subframe 0, half 0), (0, 1), (1, 0), (1, 1)
    fb.outw = DIRECT ? nullptr : (uint32_t *)(slots + (size_t)d.out_slot * P.slot_bytes) + (size_t)chunk * chunk_cap_words;
    uint32_t bitpos = 0;
    bool redo = false;

    uint32_t hl = 0;            // bytes of the frame header (wave 0)
    if (wv == 0) {   // frame header: assembled by lane 0 in LDS, emitted one byte per lane
        LDS uint8_t *hb = (LDS uint8_t *)misc;
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
        if constexpr (!DIRECT) {
            const uint32_t v = (uint32_t)lane < hl ? hb[lane] : 0, b = (uint32_t)lane < hl ? 8 : 0;
            wave_lds_fence();
            cb_or(fb, (uint32_t)lane * 8, v, b);
            bitpos = hl * 8;
            wave_lds_fence();
        }
        if (lane == 0) results[d.out_slot].ca = ca;
    }
    int32_t cca, ccb;
    uint32_t ccs;
    pipe_cand_coef(MS, c, cca, ccb, ccs);
    const uint32_t csh = (wraw & 0x100u) ? ccs : ccs + wst - pre;
    auto cand = [&](int32_t l, int32_t r) -> int32_t { return pipe_cand(l, r, cca, ccb, csh); };
    // (the walks: candidate = (a + cb b) >> shift with the rows picked per candidate, as in the evaluation -- a select fewer a sample)
    const bool cplain = !MS || c < 2;
    const int32_t cvb = cplain ? 0 : ccb;
    const LDS samp_t *const rowA = (NCH == 2 && c == 1) ? rowR : rowL, *const rowB = cplain ? rowA : rowR;
    auto cand2 = [&](int32_t a, int32_t b) __attribute__((always_inline)) -> int32_t { return (a + __mul24(b, cvb)) >> csh; };
    // (true 32-bit content, pipe_eval_cand_w32: the candidate's samples as doubles -- up to 33 bits -- instead of 24-bit integer forms)
    const bool w32 = ACC64 && P.bps == 32 && !pre_ok;
    const double wscale = (wraw & 0x100u) ? 1.0 : __hiloint2double((int)((1023u - wst) << 20), 0);
    auto candd = [&](int32_t l, int32_t r) -> double { return pipe_cdbl(l, r, c, MS, wscale); };
    // ---- everything in front of the residual: lane 0 = subframe header byte (+ the unary wasted-bits field), lanes
    // 1..order = warm-up samples, then precision/shift, coefficients, coding method + partition order.  Returns the bits; with
    // `emit` they are written from bit `at` on (DIRECT measures first and writes behind the exchange of the chunk lengths).
    uint32_t s_pv = 0, s_pb = 0, s_val = 0, s_vb = 0, s_off = 0;      // this lane's field(s) of the subframe header and where they start in it
    auto sub_fields = [&]() __attribute__((always_inline)) -> uint32_t {
        uint32_t hdr;
        switch (type) {
        case 0: hdr = 0x00; break;
        case 1: hdr = 0x02; break;
        case 2: hdr = 0x10 | (order << 1); break;
        default: hdr = 0x40 | ((order - 1) << 1); break;
        }
        uint32_t pv = 0, pb = 0, val = 0, vb = 0;
        const bool pred = type >= 2;
        const uint32_t nw = type == 0 ? 1 : (pred ? order : 0);
        if (lane == 0) {
            pv = hdr | (wst ? 1u : 0u); pb = 8;
            if (wst) { val = 1; vb = wst; }                  // wasted bits - 1 zeros, then a one
        }
        else if ((uint32_t)lane <= nw) {
            const uint32_t g = (uint32_t)lane - 1;             // sample index (inside segment 0: order <= MAXO <= seg)
            if (ACC64 && w32) {
                const i64 xi = (i64)candd(sL[g], (NCH == 2) ? (int32_t)sR[g] : 0);
                if (sb == 33) { pv = (uint32_t)((u64)xi >> 32) & 1u; pb = 1; val = (uint32_t)xi; vb = 32; }       // (a 33-bit field: its top bit, then the rest)
                else { val = (uint32_t)xi & mask; vb = sb; }
            }
            else { val = (uint32_t)cand(sL[g], (NCH == 2) ? (int32_t)sR[g] : 0) & mask; vb = sb; }
        }
        else if (type == 3 && (uint32_t)lane == order + 1) { pv = prec - 1; pb = 4; val = (uint32_t)shift & 31; vb = 5; }
        else if (type == 3 && (uint32_t)lane <= 2 * order + 1) { val = (uint32_t)dec->q[lane - order - 2] & ((1u << prec) - 1); vb = prec; }
        else if (pred && (uint32_t)lane == (type == 3 ? 2 * order + 2 : order + 1)) { val = (method << 4) | po; vb = 6; }
        const uint32_t mine = pb + vb;
        const uint32_t incl = wave_scan_add(mine);
        s_pv = pv; s_pb = pb; s_val = val; s_vb = vb; s_off = incl - mine;
        return rl(incl, 63);
    };
    auto sub_emit = [&](uint32_t at) __attribute__((always_inline)) {
        cb_or<DIRECT>(fb, at + s_off, s_pv, s_pb);
        cb_or<DIRECT>(fb, at + s_off + s_pb, s_val, s_vb);
        wave_lds_fence();
    };
    uint32_t sub_bits = 0;          // (DIRECT) bits of this wave's subframe header
    if constexpr (!DIRECT) {
        if (hf == 0) {
            const uint32_t total = sub_fields();
            cb_reserve(fb, lane, bitpos, total);
            sub_emit(bitpos);
            bitpos += total;
        }
    }
    // ---- DIRECT: the waves exchange their chunk lengths (0xFFFFFFFF: this wave hands the block back); every wave learns where
    // its chunk starts in the frame and how long the frame is.  Then the headers go in with LDS atomics -- all of them, in front of a
    // barrier: the first word of a wave's residual is read and later stored whole by its lane 0, and may hold header bits of any wave.
    uint32_t d_start = 0, d_total = 0;
    bool d_poison = false;
    auto direct_sync = [&](uint32_t body_bits, bool handback) __attribute__((always_inline)) {
        // (the fields of the subframe header -- warm-up samples among them -- are read here, in front of the barrier: with ALIAS the
        // samples are gone behind it)
        if (hf == 0) sub_bits = sub_fields();
        const uint32_t chunk_bits_ = hl * 8 + sub_bits + body_bits;
        if (lane == 0) xch[wv] = handback ? 0xFFFFFFFFu : chunk_bits_;
        __syncthreads();
        uint32_t T = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t b = xch[w];
            d_poison = d_poison || b == 0xFFFFFFFFu;
            if ((uint32_t)w < wv) d_start += b;
            T += b;
        }
        d_total = T;
        if (T + 64u > 32u * fb.fbw) d_poison = true;             // (a frame beyond the buffer: nothing of the kind comes out of 16-bit input)
        if (d_poison) { d_start = 0; d_total = 0; }
        const uint32_t nb = ((d_total + 7) >> 3) + 2;
        if (wv == 0 && lane == 0) {
            if (d_poison) {
                atomicOr(&results[d.out_slot].err, FG_ERR_REDO);
                if (B.guard) atomicOr(&B.guard[2], (unsigned long long)(FG_ERR_REDO | FG_ERR_CHAIN));
                lb_publish(D.lb, d.out_slot, lb_word(D.epoch, FG_LB_POISON, 0));
            }
            else lb_publish(D.lb, d.out_slot, lb_word(D.epoch, d.out_slot == 0 ? FG_LB_PFX : FG_LB_AGG, nb));
        }
        if constexpr (ALIAS) {
            // every wave is done with the staged samples: the frame's words take their place, zeroed first
            const uint32_t zw = ((d_total + 31) >> 5) + 2;
            for (uint32_t j = tid; j < zw; j += NT) fb.w[j] = 0;
            __syncthreads();
        }
        if (!d_poison) {
            if (wv == 0) {
                const LDS uint8_t *hb = (const LDS uint8_t *)misc;
                const uint32_t v = (uint32_t)lane < hl ? hb[lane] : 0, b = (uint32_t)lane < hl ? 8 : 0;
                wave_lds_fence();
                cb_or<true>(fb, (uint32_t)lane * 8, v, b);
                wave_lds_fence();
            }
            if (hf == 0) sub_emit(d_start + hl * 8);
        }
        __syncthreads();
    };
    uint32_t fin_cw = 0, fin_cur = 0;       // (DIRECT) the word a lane ends in and its bits there: merged behind a barrier
    bool fin_has = false;
    if (type != 0) {
        // ---- body: pass A = exact bit length of every lane's segment, prefix sum = its start, pass B = the codes
        int32_t q[MAXO];
        if (type == 3) {
            const int32_t qall = (lane < MAXO) ? dec->q[lane < 12 ? lane : 0] : 0;
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = (int32_t)rl((uint32_t)qall, j);
        }
        else {
            const uint32_t g = type == 2 ? order : 0;
            const int32_t c0 = g == 0 ? 0 : (int32_t)g, c1 = g < 2 ? 0 : (g == 2 ? -1 : g == 3 ? -3 : -6);
            const int32_t c2 = g < 3 ? 0 : (g == 3 ? 1 : 4), c3 = g < 4 ? 0 : -1;
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : j == 3 ? c3 : 0;
            shift = 0;
        }
        double qd[MAXO];
        const double scl = __hiloint2double((int)((1023u - (uint32_t)shift) << 20), 0);      // 2^-shift
        if constexpr (FGP_F64P && ACC64) {
#pragma unroll
            for (int j = 0; j < MAXO; j++) qd[j] = (double)q[j] * scl;       // (scaled: pfir_f64n)
        }
        const uint32_t plen = method ? 5 : 4;
        const uint32_t lpp = LPS >> po;                                  // lanes per partition
        const uint32_t kr = type >= 2 ? (uint32_t)dec->k[Lg / lpp] : 0;   // this lane's Rice parameter
        const bool pstart = type >= 2 && (Lg % lpp) == 0;
        const uint32_t skip = (type >= 2 && Lg == 0) ? order : 0;        // warm-up samples are not coded
        // (ALIAS: sample g of the block from memory -- the second walk of a wave that could not keep its residuals; the staged copy is gone)
        auto gcand = [&](uint32_t g) __attribute__((always_inline)) -> int32_t {
            int32_t l, r = 0;
            if (NCH == 2) {
                if (P.pcm_i16) { const short2 v = ((const short2 *)pcm)[d.pcm_off + g]; l = v.x; r = v.y; }
                else { const int2 v = ((const int2 *)pcm)[d.pcm_off + g]; l = v.x; r = v.y; }
            }
            else l = P.pcm_i16 ? (int32_t)((const int16_t *)pcm)[d.pcm_off + g] : ((const int32_t *)pcm)[d.pcm_off + g];
            return cand(l >> pre, r >> pre);
        };
        auto walk_t = [&](auto VERB, auto EMIT, auto ATOM, auto ALLF, uint32_t p0, bool inrange_) __attribute__((always_inline)) -> uint32_t {
            constexpr bool verb = decltype(VERB)::value, emit = decltype(EMIT)::value, atom = decltype(ATOM)::value;
            constexpr bool fromg = ALIAS && emit;
            const bool inrange = decltype(ALLF)::value ? true : inrange_;
            int32_t h[MAXO];
            double hd[MAXO];                    // (17..25-bit samples: pfir_f64n, as in the evaluation)
#pragma unroll
            for (int j = 0; j < MAXO; j++) {
                int32_t x = 0;
                double xdh = 0.0;
                if (Lg > 0) {
                    if (fromg) x = gcand(Lg * seg - 1u - (uint32_t)j);
                    else if (RAG) {
                        x = cand(prvL[-1 - j], (NCH == 2) ? (int32_t)prvR[-1 - j] : 0);
                        if (ACC64 && w32) xdh = candd(prvL[-1 - j], (NCH == 2) ? (int32_t)prvR[-1 - j] : 0);
                    }
                    else {
                        x = cand2(rowA[hoff - j], rowB[hoff - j]);
                        if (ACC64 && w32) xdh = candd(rowL[hoff - j], (NCH == 2) ? (int32_t)rowR[hoff - j] : 0);
                    }
                }
                h[(MAXO - 1 - j) % MAXO] = ACC64 ? ppack(x) : x;
                if constexpr (FGP_F64P && ACC64) hd[(MAXO - 1 - j) % MAXO] = w32 ? xdh : (double)x;
            }
            uint32_t pos = p0, len = 0;
            // Emission without LDS atomics: a lane's bits are consecutive, so it keeps the word it is filling in a register
            // (`cur`, window word `cw`) and stores it when it moves on; the word it ends in is shared with the next lane and
            // is merged after the walk.  The stores are unconditional: a lane that has nothing to store writes to a scratch
            // word of its own.  This needs every lane to span at least a word (no three lanes in one word): true for 32 or
            // more samples per lane; shorter segments (atom) OR their bits into the window with LDS atomics instead.
            LDS uint32_t *const dummy = misc + lane;
            uint32_t cw = (p0 >> 5) - fb.wbase;
            uint32_t cur = (emit && !atom && inrange) ? fb.w[cw] : 0;
            auto put = [&](uint32_t at, uint32_t val, uint32_t vb) __attribute__((always_inline)) {
                if (atom) { cb_or<DIRECT>(fb, inrange ? at : (fb.wbase << 5), inrange ? val : 0, vb); return; }
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
            auto step = [&](int u, uint32_t s, bool guard) __attribute__((always_inline)) {
                const int32_t x = fromg ? gcand(Lg * seg + s) : cand2(rowA[s], rowB[s]);
                uint32_t val, vb, lead, topv = 0, topb = 0;          // (topv / topb: the 33rd bit of a verbatim sample of a 33-bit side channel)
                if (verb) {
                    if (ACC64 && w32) {
                        const i64 xi = (i64)candd(rowL[s], (NCH == 2) ? (int32_t)rowR[s] : 0);
                        if (sb == 33) { topv = (uint32_t)((u64)xi >> 32) & 1u; topb = 1; val = (uint32_t)xi; vb = 32; }
                        else { val = (uint32_t)xi & mask; vb = sb; }
                    }
                    else { val = (uint32_t)x & mask; vb = sb; }
                    lead = 0;
                }
                else {
                    int32_t res;
                    if constexpr (!ACC64) res = x - (pfir24<MAXO>(q, h, u) >> shift);
                    else if constexpr (FGP_F64P) {
                        const double xd = w32 ? candd(rowL[s], (NCH == 2) ? (int32_t)rowR[s] : 0) : (double)x;
                        res = (int32_t)(-__builtin_floor(pfir_f64n<MAXO>(qd, hd, u, -xd)));
                        hd[u] = xd;
                    }
                    else res = (int32_t)((i64)x - (pfir48<MAXO>(q, h, u) >> shift));
                    h[u] = ACC64 ? ppack(x) : x;
                    const uint32_t uu = ((uint32_t)res << 1) ^ (uint32_t)(res >> 31);
                    lead = uu >> kr;
                    val = kone | (uu & kmask);
                    vb = kr + 1;
                }
                if (ACC64 && topb) {
                    if (emit) put(pos, topv, 1);
                    pos += 1; len += 1;
                }
                if (guard) {
                    const bool coded = s >= skip;
                    if (emit) put(coded ? pos + lead : pos, coded ? val : 0, coded ? vb : 0);
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
            // (ragged geometry: the first lanes of a group have one sample more than `seg`; at most MAXO steps either way)
#pragma unroll
            for (int u = 0; u < MAXO; u++) if (s0 + u < (RAG ? ln.len : seg)) step(u, s0 + u, s0 == 0);
            if (emit && !atom) {
                if constexpr (DIRECT) { fin_cw = cw; fin_cur = cur; fin_has = inrange; }
                else {
                    wave_lds_fence();
                    if (inrange) fb.w[cw] |= cur;
                    wave_lds_fence();
                }
            }
            return len;
        };
        auto walk = [&](bool emit, uint32_t p0, bool inrange, bool all) __attribute__((always_inline)) -> uint32_t {
            typedef std::integral_constant<bool, true> T;
            typedef std::integral_constant<bool, false> F;
            if (!emit) return type == 1 ? walk_t(T(), F(), F(), F(), p0, inrange) : walk_t(F(), F(), F(), F(), p0, inrange);
            if (rag || seg < 32) return type == 1 ? walk_t(T(), T(), T(), F(), p0, inrange) : walk_t(F(), T(), T(), F(), p0, inrange);
            if (all) return type == 1 ? walk_t(T(), T(), F(), T(), p0, true) : walk_t(F(), T(), F(), T(), p0, true);
            return type == 1 ? walk_t(T(), T(), F(), F(), p0, inrange) : walk_t(F(), T(), F(), F(), p0, inrange);
        };
        // ---- KEEP: the measuring walk with the residuals kept (uu = zig-zagged residual, 0 for the warm-up samples lane 0 does not code)
        uint32_t pk[16];
        bool keep_ok = false;
        auto walk_keep = [&]() __attribute__((always_inline)) -> uint32_t {
            int32_t h[MAXO];
#pragma unroll
            for (int j = 0; j < MAXO; j++) {
                int32_t x = 0;
                if (Lg > 0) x = cand2(rowA[hoff - j], rowB[hoff - j]);
                h[(MAXO - 1 - j) % MAXO] = x;
            }
            uint32_t len = pstart ? plen : 0u, big = 0;
#pragma unroll
            for (int s = 0; s < 32; s++) {
                const int u = s % MAXO;
                const int32_t x = cand2(rowA[s], rowB[s]);
                const int32_t res = x - (pfir24<MAXO>(q, h, u) >> shift);
                h[u] = x;
                uint32_t uu = ((uint32_t)res << 1) ^ (uint32_t)(res >> 31);
                const bool coded = s >= MAXO || (uint32_t)s >= skip;
                big |= coded ? uu >> 16 : 0u;
                uu = coded ? uu : 0u;
                len += coded ? (uu >> kr) + kr + 1 : 0u;
                if (s & 1) pk[s >> 1] |= uu << 16; else pk[s >> 1] = uu & 0xFFFFu;
                // (left alone the scheduler requests all 64 LDS reads of the unrolled loop up front and spills eighty registers)
                if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            keep_ok = !__any(big != 0);
            return len;
        };
        // ---- KEEP: the codes from the kept residuals (all 64 lanes at once: every lane's bits fit the window)
        auto emit_keep = [&](uint32_t p0) __attribute__((always_inline)) {
            LDS uint32_t *const dummy = misc + lane;
            uint32_t cw = (p0 >> 5) - fb.wbase;
            uint32_t cur = fb.w[cw];
            uint32_t pos = p0;
            auto put = [&](uint32_t at, uint32_t val, uint32_t vb) __attribute__((always_inline)) {
                const uint32_t rel = at - (fb.wbase << 5);
                const uint32_t wi = rel >> 5, sh = rel & 31;
                const u64 x = (u64)val << ((64 - sh - vb) & 63);
                const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
                const bool moved = wi != cw;
                *(moved ? fb.w + cw : dummy) = cur;
                cur = moved ? hi : (cur | hi);
                cw = wi;
                const bool spill = lo != 0;
                *(spill ? fb.w + cw : dummy) = cur;
                cur = spill ? lo : cur;
                cw += spill ? 1u : 0u;
            };
            if (pstart) { put(pos, kr, plen); pos += plen; }
            const uint32_t kmask = (1u << kr) - 1, kone = 1u << kr;
#pragma unroll
            for (int s = 0; s < 32; s++) {
                const uint32_t uu = (s & 1) ? pk[s >> 1] >> 16 : pk[s >> 1] & 0xFFFFu;
                const uint32_t lead = uu >> kr, val = kone | (uu & kmask), vb = kr + 1;
                if (s < MAXO) {
                    const bool coded = (uint32_t)s >= skip;
                    put(coded ? pos + lead : pos, coded ? val : 0, coded ? vb : 0);
                    pos += coded ? lead + vb : 0u;
                }
                else { put(pos + lead, val, vb); pos += lead + vb; }
                if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DIRECT) { fin_cw = cw; fin_cur = cur; fin_has = true; }
            else {
                wave_lds_fence();
                fb.w[cw] |= cur;
                wave_lds_fence();
            }
        };
        const bool use_keep = KEEP && !ACC64 && !RAG && WS == 2 && ws == 2 && seg == 32 && type >= 2;
        uint32_t mylen0;
        if (KEEP && use_keep) mylen0 = walk_keep();
        else mylen0 = walk(false, 0, false, false);
        const uint32_t mylen = ln.act ? mylen0 : 0u;                 // (idle lanes of the ragged geometry code nothing)
        if (__any(mylen > (1u << 24))) redo = true;                  // absurd code lengths: the generic kernel copes
        if constexpr (DIRECT) {
            const uint32_t incl = wave_scan_add(redo ? 0u : mylen);
            const uint32_t body = rl(incl, 63);
            direct_sync(body, redo);
            if (!d_poison) {
                const uint32_t mystart = d_start + hl * 8 + sub_bits + incl - mylen;
                if (KEEP && use_keep && keep_ok) emit_keep(mystart);
                else (void)walk(true, mystart, true, true);
            }
            bitpos = hl * 8 + sub_bits + body;
        }
        else if (!redo) {
            const uint32_t incl = wave_scan_add(mylen);
            const uint32_t mystart = bitpos + incl - mylen, myend = bitpos + incl;
            const uint32_t subend = bitpos + rl(incl, 63);
            uint32_t a = 0;
            if (KEEP && use_keep && keep_ok) {
                // (outside the loop below: the kept residuals must not stay alive through the two-walk form's code)
                cb_flush(fb, lane, rl(mystart, 0), false);
                const uint32_t cap = (fb.wbase << 5) + 32u * fb.fbw - 64u;
                if (__ballot(myend <= cap) == ~0ull) { emit_keep(mystart); wave_lds_fence(); a = 64; }
            }
#pragma unroll 1
            while (a < 64) {
                cb_flush(fb, lane, rl(mystart, (int)a), false);
                const uint32_t cap = (fb.wbase << 5) + 32u * fb.fbw - 64u;
                const uint64_t fits = __ballot((uint32_t)lane >= a && myend <= cap);
                const uint64_t shifted = fits >> a;
                const uint32_t cnt = (~shifted) ? (uint32_t)__builtin_ctzll(~shifted) : 64u - a;
                if (cnt == 0) { redo = true; break; }
                const uint32_t b = a + cnt;
                (void)walk(true, mystart, ln.act && (uint32_t)lane >= a && (uint32_t)lane < b, a == 0 && b == 64 && !rag);
                wave_lds_fence();
                a = b;
            }
            bitpos = subend;
        }
    }
    else if constexpr (DIRECT) {
        direct_sync(0, false);
        bitpos = hl * 8 + sub_bits;
    }
    if constexpr (DIRECT) {
        __syncthreads();                    // every whole word of the residual walks is stored: now the words two lanes (or waves) share
        if (fin_has && !d_poison) __hip_atomic_fetch_or(&fb.w[fin_cw], fin_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0) B.chunk_bits[(size_t)d.out_slot * 4 + chunk] = d_poison ? 0 : bitpos;
        const uint32_t nbytes = (d_total + 7) >> 3, nb = nbytes + 2;            // frame without / with its CRC-16
        // where the frame goes: the sizes in front of it.  Wave 0 sends for them now and looks at them behind the CRC pass
        u64 lbv[FG_LB_NWIN];
        const bool look = wv == 0 && !d_poison && d.out_slot != 0 && !(D.reserved & 1u);
        if (look) lb_fetch(D.lb, (int64_t)d.out_slot - 1, D.epoch, lane, lbv);
        __syncthreads();
        // CRC-16: thread t owns the 16-byte granules t, t + NT, ... (zero granules in front so that the last step is full):
        // state * x^(128 NT) + crc(granule), everything through look-up tables in LDS (the closed form of the polynomial costs forty
        // instructions a word, four look-ups seventeen): ctab [0,512) the multiplication, [512,1536) a byte followed by 3, 2, 1, 0 zero
        // bytes.  Folded at the end with x^(128 (NT - 1 - t) + 8 rem), rem = the bytes behind the last whole granule, which the last
        // thread takes meanwhile.  Words in the window are most significant bit first.
        const LDS uint16_t *T3 = ctab + 512, *T2 = ctab + 768, *T1 = ctab + 1024, *T0 = ctab + 1280;
        auto crcw = [&](uint32_t c, uint32_t w) __attribute__((always_inline)) -> uint32_t {
            if (!FGX_CRCTAB) return crc16_word(c, w);
            return (uint32_t)T3[(c >> 8) ^ (w >> 24)] ^ (uint32_t)T2[(c & 0xFF) ^ ((w >> 16) & 0xFF)] ^ (uint32_t)T1[(w >> 8) & 0xFF] ^ (uint32_t)T0[w & 0xFF];
        };
        const uint32_t W = nbytes >> 2, tail = nbytes & 3, G = W >> 2, Wr = W & 3, rem = nbytes & 15;
        if (!d_poison && !(D.reserved & 2u)) {
            const uint32_t pad = (NT - (G % NT)) % NT, Tn = (G + pad) / NT;
            const uint32_t foldc = D.crcx[1536 + rem * NT + tid];
            uint32_t st = 0;
            for (uint32_t t = 0; t < Tn; t++) {
                const int qi = (int)(t * NT + (uint32_t)tid) - (int)pad;
                typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
                u32x4 g = {0, 0, 0, 0};
                if (qi >= 0) g = *(const LDS u32x4 *)(fb.w + 4 * qi);
                st = (uint32_t)ctab[st >> 8] ^ (uint32_t)ctab[256 + (st & 0xFF)];
                uint32_t cc = crcw(0, g.x);
                cc = crcw(cc, g.y); cc = crcw(cc, g.z); cc = crcw(cc, g.w);
                st ^= cc;
            }
            if (Tn) st = gf16_mul(st, foldc);
            if (tid == NT - 1) {
                uint32_t cr = 0;
                for (uint32_t k = 0; k < Wr; k++) cr = crcw(cr, fb.w[4 * G + k]);
                const uint32_t wvl = fb.w[W];
                for (uint32_t b = 0; b < tail; b++) cr = ((cr << 8) & 0xFFFF) ^ (uint32_t)T0[((cr >> 8) ^ (wvl >> (24 - 8 * b))) & 0xFF];
                st ^= cr;
            }
            st = wave_xor32(st);
            if (lane == 0) __hip_atomic_fetch_xor(&xch[12], st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (wv == 0) {
            u64 excl = 0;
            bool ok = !d_poison;
            if (D.reserved & 1u) excl = (u64)d.out_slot * 11000ull;          // (tuning builds, FLACGPU_DIRECT_X: timing experiments, wrong places)
            else if (look) {
                ok = lb_finish(D.lb, D.epoch, lane, lbv, (int64_t)d.out_slot - 1, excl, true);
                if (lane == 0) {
                    if (ok) lb_publish(D.lb, d.out_slot, lb_word(D.epoch, FG_LB_PFX, excl + nb));
                    else {
                        lb_publish(D.lb, d.out_slot, lb_word(D.epoch, FG_LB_POISON, 0));
                        if (B.guard) atomicOr(&B.guard[2], (unsigned long long)FG_ERR_CHAIN);
                    }
                }
            }
            if (lane == 0) { xch[8] = (uint32_t)excl; xch[9] = (uint32_t)(excl >> 32); xch[10] = ok ? 1u : 0u; }
        }
        __syncthreads();
        if (xch[10] == 0 || (D.reserved & 2u)) return;           // (the whole workgroup: the call falls back to the chunk form)
        const u64 excl = (u64)xch[8] | ((u64)xch[9] << 32);
        if (excl + nb > D.dst_cap) {                                 // (the host reports the short buffer once it has read the total)
            if (tid == 0) {
                results[d.out_slot].bytes = nb;
                D.offsets[d.out_slot] = excl;
                if (D.user_offsets) D.user_offsets[d.out_slot] = excl;
                if (d.out_slot + 1 == D.nblocks) { D.offsets[D.nblocks] = excl + nb; if (D.user_offsets) D.user_offsets[D.nblocks] = excl + nb; }
            }
            return;
        }
        uint8_t *out = D.dst + excl;
        struct __attribute__((packed)) U32 { uint32_t v; };
        // the bytes: consecutive lanes store consecutive words -- 256 contiguous bytes an instruction, wherever the frame starts
        for (uint32_t j = (uint32_t)tid; j < W; j += NT) ((U32 *)out)[j].v = __builtin_bswap32(fb.w[j]);
        if (tid == NT - 1) {
            const uint32_t wvl = fb.w[W];
            for (uint32_t b = 0; b < tail; b++) out[W * 4 + b] = (uint8_t)(wvl >> (24 - 8 * b));
        }
        if (tid == 0) {
            const uint32_t crc = xch[12];
            out[nbytes] = (uint8_t)(crc >> 8); out[nbytes + 1] = (uint8_t)crc;
            results[d.out_slot].bytes = nb;
            D.offsets[d.out_slot] = excl;
            if (D.user_offsets) D.user_offsets[d.out_slot] = excl;
            if (d.out_slot + 1 == D.nblocks) { D.offsets[D.nblocks] = excl + nb; if (D.user_offsets) D.user_offsets[D.nblocks] = excl + nb; }
        }
        return;
    }
    if (!redo) cb_flush(fb, lane, bitpos, true);
    if (lane == 0) {
        B.chunk_bits[(size_t)d.out_slot * 4 + chunk] = redo ? 0 : bitpos;
        const uint32_t e = fb.err | (redo ? FG_ERR_REDO : 0u);
        if (e) { atomicOr(&results[d.out_slot].err, e); if (B.guard) atomicOr(&B.guard[2], (unsigned long long)e); }
    }
}

// ================================================================================================ K6: chunks -> frame at its final place
// One wave per block.  Pipeline blocks (results.reserved == 4): the frame's words are gathered from the chunks with funnel
// shifts, the CRC-16 runs alongside (lane l owns the words l, l+64, ...: state * x^2048 + crc(word) through tables in LDS,
// final fold with x^(32k) multipliers), and the bytes go to dst + offsets[slot].  Other blocks (generic kernel: complete
// frames in their slots) are copied.
template <int WPB>
__global__ void __launch_bounds__(WPB * 64)
fg_pipe_assemble_kernel(const FgBlockDesc *descs, uint32_t nblocks, const uint8_t *slots, uint32_t slot_bytes, uint32_t chunk_cap_words,
                        uint32_t nw, const uint32_t *chunk_bits, FgBlockResult *results, u64 *offsets, uint8_t *dst,
                        u64 dst_cap, const uint16_t *crctab, u64 *user_offsets, const unsigned long long *guard, uint32_t first, FgPackDirect D)
{
    __shared__ uint16_t tab[1792];           // [0,256) byte table, [256,768) x^2048 tables, [768,832) x^(32k), [1024,1792) slicing tables
    for (uint32_t j = threadIdx.x; j < 1792; j += WPB * 64) tab[j] = crctab[j];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // (behind a direct launch, D.lb set: the blocks [first, nblocks) of the list that kept the chunk form; their places come from
    // the look-back words -- everything is published by now --, and the first wave hands the error flags to the host's words)
    const uint32_t bi = first + blockIdx.x * WPB + (threadIdx.x >> 6);
    if (D.lb && lane == 0 && bi == first) offsets[D.nblocks + 1] = guard ? guard[2] : 0ull;
    if (bi >= nblocks) return;
    const FgBlockDesc d = descs[bi];
    const FgBlockResult res = results[d.out_slot];
    // chunk table of a pipeline block (wave-uniform); its size is ceil(sum of the chunk bits / 8) + 2 for the CRC-16, and 0
    // while it waits for the generic kernel (FG_ERR_REDO)
    uint32_t S[4], Bc[4];
    const uint32_t *cw[4];
    uint32_t T = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t b = ((uint32_t)w < nw && res.reserved == 4) ? chunk_bits[(size_t)d.out_slot * 4 + w] : 0;
        S[w] = T; Bc[w] = b; T += b;
        cw[w] = (const uint32_t *)(slots + (size_t)d.out_slot * slot_bytes) + (size_t)w * chunk_cap_words;
    }
    const uint32_t nb = res.reserved == 4 ? ((res.err & FG_ERR_REDO) ? 0u : ((T + 7) >> 3) + 2) : res.bytes;
    if (lane == 0 && res.reserved == 4) results[d.out_slot].bytes = nb;        // (the scan no longer writes the pipeline's sizes back)
    u64 myoff;
    if (D.lb) {
        if (!lb_lookback(D.lb, d.out_slot, D.epoch, lane, myoff, false)) return;       // (a broken chain: the direct kernels have said so, FG_ERR_CHAIN)
        if (lane == 0) {
            offsets[d.out_slot] = myoff;
            if (user_offsets) user_offsets[d.out_slot] = myoff;
            if (d.out_slot + 1 == D.nblocks) { offsets[D.nblocks] = myoff + nb; if (user_offsets) user_offsets[D.nblocks] = myoff + nb; }
        }
    }
    else {
        myoff = offsets[d.out_slot];
        // the frame index for the caller (saves a device-to-device copy), and the guard counters beside the totals the host reads
        if (lane == 0 && user_offsets) {
            user_offsets[d.out_slot] = myoff;
            if (bi == 0) user_offsets[nblocks] = offsets[nblocks];
        }
        if (lane == 0 && bi == 0 && guard) { offsets[nblocks + 2] = guard[0]; offsets[nblocks + 3] = guard[1]; }
    }
    if (nb == 0 || myoff + nb > dst_cap) return;      // the host reports the short buffer once it has read the total
    uint8_t *out = dst + myoff;
    struct __attribute__((packed)) U32 { uint32_t v; };
    if (res.reserved != 4) {
        const uint32_t *sw = (const uint32_t *)(slots + (size_t)d.out_slot * slot_bytes);
        const uint32_t nwd = nb >> 2;
        for (uint32_t j = lane; j < nwd; j += 64) ((U32 *)out)[j].v = sw[j];
        const uint32_t done = nwd * 4;
        if ((uint32_t)lane < nb - done) out[done + lane] = ((const uint8_t *)sw)[done + lane];
        return;
    }
    const uint32_t nbytes = (T + 7) >> 3;            // frame without its CRC-16
    const uint32_t W = nbytes >> 2, tail = nbytes & 3;
    auto gather = [&](uint32_t j) -> uint32_t {
        uint32_t v = 0;
        const int64_t b0 = (int64_t)j * 32;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int64_t rel = b0 - (int64_t)S[w];
            if (Bc[w] != 0 && rel > -32 && rel < (int64_t)Bc[w]) {       // (an empty chunk's words were never written)
                uint32_t x;
                if (rel >= 0) {
                    const uint32_t wi = (uint32_t)rel >> 5, sh = (uint32_t)rel & 31;
                    const uint32_t hi = cw[w][wi];
                    const uint32_t lo = (((wi + 1) << 5) < Bc[w]) ? cw[w][wi + 1] : 0;
                    x = (uint32_t)((((u64)hi << 32) | lo) >> (32 - sh));
                }
                else x = cw[w][0] >> (uint32_t)(-rel);
                v |= x;
            }
        }
        return v;
    };
    uint32_t crc = 0;
    const uint32_t nrows = (W + 63) >> 6;
    auto emit = [&](uint32_t j, uint32_t v) __attribute__((always_inline)) {
        if (j < W) {
            ((U32 *)out)[j].v = __builtin_bswap32(v);
            uint32_t s = crc;
            s = tab[256 + (s >> 8)] ^ tab[512 + (s & 0xFF)];
            crc = s ^ crc16_word(0, v);           // (closed form: no look-ups at data-dependent addresses)
        }
    };
    // the chunk a run of rows [row, row + cnt) lies inside entirely (every word of it two loads and a funnel shift), or -1
    auto interior = [&](uint32_t row, uint32_t cnt) -> int {
        const uint32_t rb0 = row * 2048, rb1 = (row + cnt) * 2048 + 32;
        int in = -1;
#pragma unroll
        for (int w = 0; w < 4; w++) if (rb0 >= S[w] && rb1 <= S[w] + Bc[w]) in = w;
        return in;
    };
    uint32_t row = 0;
    // Four rows per step while they lie inside one chunk (all but a handful do): eight loads per lane, and the loads of the
    // NEXT group are requested before the current one is turned into frame words, so that a frame's ~12 groups do not each
    // wait out a trip to memory.
    uint32_t a[4], b[4], sh = 0;
    auto issue = [&](uint32_t r0, int in_, uint32_t (&xa)[4], uint32_t (&xb)[4], uint32_t &xsh) __attribute__((always_inline)) {
        const uint32_t *p = in_ == 0 ? cw[0] : in_ == 1 ? cw[1] : in_ == 2 ? cw[2] : cw[3];
        const uint32_t s_ = in_ == 0 ? S[0] : in_ == 1 ? S[1] : in_ == 2 ? S[2] : S[3];
        const uint32_t rel = (r0 * 64 + (uint32_t)lane) * 32 - s_;
        const uint32_t wi = rel >> 5;
        xsh = rel & 31;
#pragma unroll
        for (int u = 0; u < 4; u++) { xa[u] = p[wi + 64 * u]; xb[u] = p[wi + 64 * u + 1]; }
    };
    int in = (row + 4 <= nrows) ? interior(row, 4) : -1;
    if (in >= 0) issue(row, in, a, b, sh);
    while (row + 4 <= nrows) {
        if (in < 0) {
            // a chunk boundary inside these rows: one row the general way, then try again
            const uint32_t j = row * 64 + (uint32_t)lane;
            emit(j, gather(j));
            row++;
            in = (row + 4 <= nrows) ? interior(row, 4) : -1;
            if (in >= 0) issue(row, in, a, b, sh);
            continue;
        }
        const uint32_t nrow = row + 4;
        const int nin = (nrow + 4 <= nrows) ? interior(nrow, 4) : -1;
        uint32_t na[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0}, nsh = 0;
        if (nin >= 0) issue(nrow, nin, na, nb, nsh);
        const uint32_t j0 = row * 64 + (uint32_t)lane;
#pragma unroll
        for (int u = 0; u < 4; u++) emit(j0 + 64 * u, (uint32_t)((((u64)a[u] << 32) | b[u]) >> (32 - sh)));
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = na[u]; b[u] = nb[u]; }
        sh = nsh; row = nrow; in = nin;
    }
    for (; row < nrows; row++) {
        const uint32_t j = row * 64 + (uint32_t)lane;
        const int in = interior(row, 1);
        uint32_t v;
        if (in >= 0) {
            const uint32_t *p = in == 0 ? cw[0] : in == 1 ? cw[1] : in == 2 ? cw[2] : cw[3];
            const uint32_t s_ = in == 0 ? S[0] : in == 1 ? S[1] : in == 2 ? S[2] : S[3];
            const uint32_t rel = j * 32 - s_;
            const uint32_t wi = rel >> 5, sh = rel & 31;
            v = (uint32_t)((((u64)p[wi] << 32) | p[wi + 1]) >> (32 - sh));
        }
        else v = gather(j);
        emit(j, v);
    }
    // fold: lane l's last word is dist = (W - 1 - l) mod 64 words from the end
    uint32_t s = 0;
    if ((uint32_t)lane < W) s = gf16_mul(crc, tab[768 + ((W - 1 - (uint32_t)lane) & 63)]);
    uint32_t c16 = wave_xor32(s);
    if (lane == 0) {
        const uint32_t wvl = tail ? gather(W) : 0;
        for (uint32_t b = 0; b < tail; b++) {
            const uint32_t byte = (wvl >> (24 - 8 * b)) & 0xFF;
            out[W * 4 + b] = (uint8_t)byte;
            c16 = ((c16 << 8) & 0xFFFF) ^ tab[((c16 >> 8) ^ byte) & 0xFF];
        }
        out[nbytes] = (uint8_t)(c16 >> 8);
        out[nbytes + 1] = (uint8_t)c16;
    }
}

}  // namespace
