// flac_enc_fast_impl.h -- specialised one-block-per-wavefront FLAC frame encoder for the common shapes:
// 1 or 2 channels, block length a multiple of 64 whose finest Rice partition is a multiple of 64 samples
// (4096 at partition order <= 6), max LPC order <= 12, bits-per-sample <= 24.  Everything else goes to
// the generic kernel in flac_enc_kernels.hip; both produce identical bytes (tests/test_gpu_encode.py).
//
// Same algorithm and stage order as the generic kernel; what changes is the mapping:
//   * compile-time candidate set (L,R,M,S / L,R / mono) -- no per-sample branches in the analysis passes;
//   * every pass handles all candidates at once from one set of LDS loads;
//   * fixed and LPC predictors share one FIR evaluation pass (a fixed predictor of order k is the FIR with
//     binomial coefficients and shift 0), coefficients live in SGPRs;
//   * reductions and prefix sums use DPP row shifts / broadcasts instead of LDS shuffles;
//   * Rice partition sums, parameters and bit estimates live in registers, lane = partition;
//   * LDS accesses of the single wave are ordered by issue, so stages are separated by compiler fences only;
//   * wave-uniform state (bit position, decisions) is kept in plain locals so it stays in SGPRs and every
//     branch on it is a scalar branch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fg_dev.h"
#include "fg_types.h"

#define FG_LN2 0.69314718055994530942
#define FG_PADF 32   // zero samples kept in front of each staged channel (>= max order)
// LDS arrays are addressed through address_space(3) pointers: ds_* instructions, and no FLAT aperture checks
// on the (possibly negative) sample offsets.
#define LDS __attribute__((address_space(3)))
#define FGI __device__ __forceinline__

using namespace fgdev;

namespace {

template <bool ACC64> struct FastTypes {
    typedef typename std::conditional<ACC64, u64, uint32_t>::type sum_t;
    // bits-per-sample <= 16 (the !ACC64 shapes): samples are staged as int16, halving the LDS footprint
    typedef typename std::conditional<ACC64, int32_t, int16_t>::type samp_t;
};

// Constants of one block (written once).
template <bool ACC64> struct FastCtx {
    typedef typename FastTypes<ACC64>::samp_t samp_t;
    const LDS samp_t *pl, *pr;   // staged samples (after the zero padding)
    LDS double *dbuf;
    LDS double *autoc;
    LDS int32_t *qres;
    LDS uint32_t *lres;
    LDS int32_t *bestq;
    LDS uint32_t *win;
    LDS uint16_t *crct;
    LDS uint32_t *misc;
    const float *window;
    int lane;
    uint32_t n, nvec;
};

// Bit writer state (wave-uniform).
struct FastBW {
    uint32_t bitpos, wbase, err, slot_words;
    uint32_t *outw;
};

template <bool MS, int C> FGI int32_t fcv(int32_t L, int32_t R, uint32_t w)
{
    if (!MS) return (C == 0 ? L : R) >> w;
    if (C == 0) return L >> w;
    if (C == 1) return R >> w;
    if (C == 2) return ((L + R) >> 1) >> w;
    return (L - R) >> w;
}
// runtime (wave-uniform) candidate index
template <bool MS> FGI int32_t fcv_rt(uint32_t c, int32_t L, int32_t R, uint32_t w)
{
    int32_t v;
    if (!MS) v = c == 0 ? L : R;
    else v = c == 0 ? L : c == 1 ? R : c == 2 ? ((L + R) >> 1) : (L - R);
    return v >> w;
}

// ------------------------------------------------------------------ bit writer (LDS window -> HBM slot)
FGI void bw_flush(FastBW &b, LDS uint32_t *win, int lane, uint32_t newpos)
{
    const uint32_t nfull = (newpos >> 5) - b.wbase;
    if (nfull == 0) return;
    if (b.wbase + nfull > b.slot_words) { b.err |= FG_ERR_SLOT; b.wbase += nfull; return; }
    if (nfull < 64) {
        const uint32_t v = win[lane];
        if ((uint32_t)lane < nfull) b.outw[b.wbase + lane] = __builtin_bswap32(v);
        const uint32_t carry = rl(v, (int)nfull);
        wave_lds_fence();
        if ((uint32_t)lane <= nfull) win[lane] = (lane == 0) ? carry : 0;
    }
    else {
        for (uint32_t j = lane; j < nfull; j += 64) b.outw[b.wbase + j] = __builtin_bswap32(win[j]);
        const uint32_t carry = win[nfull];
        wave_lds_fence();
        for (uint32_t j = lane; j < FG_WINW + 2; j += 64) win[j] = 0;
        wave_lds_fence();
        if (lane == 0) win[0] = carry;
    }
    b.wbase += nfull;
    wave_lds_fence();
}
FGI void bw_or(const FastBW &b, LDS uint32_t *win, uint32_t pos, uint32_t val, uint32_t vbits)
{
    const uint32_t rel = pos - (b.wbase << 5);
    const uint32_t word = rel >> 5, sh = rel & 31;
    const u64 x = (u64)val << (64 - sh - vbits);
    __hip_atomic_fetch_or(&win[word], (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const uint32_t lo = (uint32_t)x;
    if (lo) __hip_atomic_fetch_or(&win[word + 1], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// rare: a round that does not fit the window (very long unary runs); serialise the lanes
FGI void bw_round_slow(FastBW &b, LDS uint32_t *win, int lane, uint32_t pv, uint32_t pb, uint32_t val, uint32_t vb, uint32_t nb)
{
    for (int L = 0; L < 64; L++) {
        const uint32_t lpv = rl(pv, L), lpb = rl(pb, L), lval = rl(val, L), lvb = rl(vb, L), lnb = rl(nb, L);
        if (lpb) {
            if (lane == 0) bw_or(b, win, b.bitpos, lpv, lpb);
            wave_lds_fence();
            b.bitpos += lpb;
            bw_flush(b, win, lane, b.bitpos);
        }
        if (lnb) {
            uint32_t z = lnb - lvb;
            while (((b.bitpos + z) >> 5) - b.wbase >= FG_WINW) {
                const uint32_t np = (b.wbase + FG_WINW) << 5;
                z -= np - b.bitpos;
                b.bitpos = np;
                bw_flush(b, win, lane, np);
            }
            b.bitpos += z;
            bw_flush(b, win, lane, b.bitpos);
            if (lvb) {
                if (lane == 0) bw_or(b, win, b.bitpos, lval, lvb);
                wave_lds_fence();
                b.bitpos += lvb;
                bw_flush(b, win, lane, b.bitpos);
            }
        }
    }
}
// One packing round: every lane may contribute a prefix field (pv, pb bits) followed by a code of nb bits whose
// low vb bits are val and whose leading nb - vb bits are zero (pb, vb <= 32).
FGI void bw_round(FastBW &b, LDS uint32_t *win, int lane, uint32_t pv, uint32_t pb, uint32_t val, uint32_t vb, uint32_t nb)
{
    const uint32_t mine = pb + nb;
    const uint32_t incl = wave_scan_add(mine);
    const uint32_t total = rl(incl, 63);
    if (total == 0) return;
    const bool anybig = __any(nb > (1u << 20));
    if (!anybig && (b.bitpos & 31) + total <= 32u * FG_WINW) {
        const uint32_t off = b.bitpos + incl - mine;
        if (pb) bw_or(b, win, off, pv, pb);
        if (vb) bw_or(b, win, off + pb + nb - vb, val, vb);
        wave_lds_fence();
        b.bitpos += total;
        bw_flush(b, win, lane, b.bitpos);
    }
    else bw_round_slow(b, win, lane, pv, pb, val, vb, nb);
}
FGI void bw_put(FastBW &b, LDS uint32_t *win, int lane, uint32_t val, uint32_t bits)
{
    bw_round(b, win, lane, 0, 0, lane == 0 ? (bits < 32 ? (val & ((1u << bits) - 1)) : val) : 0, lane == 0 ? bits : 0, lane == 0 ? bits : 0);
}
FGI void bw_flush_all(FastBW &b, LDS uint32_t *win, int lane)
{
    bw_flush(b, win, lane, b.bitpos);
    if ((b.bitpos & 31) && lane == 0) {
        if (b.wbase < b.slot_words) b.outw[b.wbase] = __builtin_bswap32(win[0]);
    }
    if ((b.bitpos & 31) && b.wbase >= b.slot_words) b.err |= FG_ERR_SLOT;
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

// ------------------------------------------------------------------ the kernel
template <bool MS, int NCH, int MAXO, bool ACC64>
__global__ void __launch_bounds__(64)
fg_encode_fast_kernel(const void *pcm, const FgBlockDesc *descs, const float *windows, FgEncParams P, uint8_t *out,
                      FgBlockResult *results, FgDebugRec *dbg, const uint16_t *crctab)
{
    constexpr int NC = MS ? 4 : NCH;
    typedef typename FastTypes<ACC64>::sum_t sum_t;
    typedef typename FastTypes<ACC64>::samp_t samp_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FgBlockDesc d = descs[blockIdx.x];
    const int lane = threadIdx.x;
    const uint32_t n = d.n;
    // ---- LDS carve
    LDS unsigned char *lbase = (LDS unsigned char *)smem;
    uint32_t off = 0;
#define FG_CARVE(type, bytes) (LDS type *)(lbase + off); off += (uint32_t)(((bytes) + 15) & ~15u)
    LDS samp_t *sL = FG_CARVE(samp_t, (P.sig_stride + FG_PADF) * sizeof(samp_t));
    LDS samp_t *sR = FG_CARVE(samp_t, NCH == 2 ? (P.sig_stride + FG_PADF) * sizeof(samp_t) : 16);
    FastCtx<ACC64> k;
    k.pl = sL + FG_PADF; k.pr = sR + FG_PADF;
    k.dbuf = FG_CARVE(double, P.lds_dbuf_bytes);
    k.autoc = FG_CARVE(double, NC * P.nvec * (MAXO + 1) * 8);
    k.qres = FG_CARVE(int32_t, NC * P.nvec * MAXO * 4);
    k.lres = FG_CARVE(uint32_t, NC * P.nvec * 4);
    k.bestq = FG_CARVE(int32_t, NC * MAXO * 4);
    k.win = FG_CARVE(uint32_t, (FG_WINW + 2) * 4);
    k.crct = FG_CARVE(uint16_t, 768 * 2);
    k.misc = FG_CARVE(uint32_t, 128 * 4);
#undef FG_CARVE
    k.window = windows + d.win_off;
    k.lane = lane; k.n = n; k.nvec = P.nvec;
    for (int j = lane; j < 768; j += 64) k.crct[j] = crctab[j];
    k.misc[64 + lane] = crctab[768 + lane];
    FgDebugRec *mydbg = dbg ? dbg + d.out_slot : nullptr;
#define FG_STAMP(i) do { if (mydbg && lane == 0) mydbg->t[i] = clock64(); } while (0)
    FG_STAMP(0);
    uint32_t err = 0;

    // ================================================================ stage: HBM -> LDS (coalesced)
    {
        const int32_t lim = (int32_t)(P.bps - 1);
        uint32_t bad = 0;
        if (lane < FG_PADF) { sL[lane] = 0; if (NCH == 2) sR[lane] = 0; }
        LDS samp_t *dl = sL + FG_PADF, *dr = sR + FG_PADF;
        for (uint32_t i0 = 0; i0 < n; i0 += 256) {
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
                    dl[i] = (samp_t)a[u];
                    if (NCH == 2) dr[i] = (samp_t)b[u];
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
    uint32_t wst[NC], sbp[NC];
    u64 tot[NC][5];
    {
        sum_t acc[NC][5];
        uint32_t orv[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            orv[c] = 0;
#pragma unroll
            for (int kk = 0; kk < 5; kk++) acc[c][kk] = 0;
        }
        for (uint32_t i = lane; i < n; i += 64) {
            int32_t l[5], r[5];
#pragma unroll
            for (int kk = 0; kk < 5; kk++) { l[kk] = k.pl[(int)i - kk]; r[kk] = (NCH == 2) ? k.pr[(int)i - kk] : 0; }
            const bool on = i >= 4;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                int32_t v[5];
#pragma unroll
                for (int kk = 0; kk < 5; kk++) {
                    if (!MS) v[kk] = (c == 0) ? l[kk] : r[kk];
                    else v[kk] = (c == 0) ? l[kk] : (c == 1) ? r[kk] : (c == 2) ? ((l[kk] + r[kk]) >> 1) : (l[kk] - r[kk]);
                }
                orv[c] |= (uint32_t)v[0];
                const int32_t e1 = v[0] - v[1], d1 = v[1] - v[2], d2 = v[2] - v[3], d3 = v[3] - v[4];
                const int32_t e2 = e1 - d1, f2 = d1 - d2, g2 = d2 - d3;
                const int32_t e3 = e2 - f2, f3 = f2 - g2;
                const int32_t e4 = e3 - f3;
                if (on) {
                    acc[c][0] += (uint32_t)abs(v[0]); acc[c][1] += (uint32_t)abs(e1); acc[c][2] += (uint32_t)abs(e2);
                    acc[c][3] += (uint32_t)abs(e3); acc[c][4] += (uint32_t)abs(e4);
                }
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
            for (int kk = 0; kk < 5; kk++) tot[c][kk] = ACC64 ? wave_sum64((u64)acc[c][kk]) : (u64)wave_sum((uint32_t)acc[c][kk]);
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
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (false) {
                u64 a5[5] = {0, 0, 0, 0, 0};
                for (uint32_t i = 4 + lane; i < n; i += 64) {
                    int32_t v[5];
#pragma unroll
                    for (int kk = 0; kk < 5; kk++) v[kk] = fcv_rt<MS>((uint32_t)c, k.pl[(int)i - kk], (NCH == 2) ? k.pr[(int)i - kk] : 0, wst[c]);
                    const int32_t e1 = v[0] - v[1], d1 = v[1] - v[2], d2 = v[2] - v[3], d3 = v[3] - v[4];
                    const int32_t e2 = e1 - d1, f2 = d1 - d2, g2 = d2 - d3;
                    const int32_t e3 = e2 - f2, f3 = f2 - g2;
                    const int32_t e4 = e3 - f3;
                    a5[0] += (uint32_t)abs(v[0]); a5[1] += (uint32_t)abs(e1); a5[2] += (uint32_t)abs(e2);
                    a5[3] += (uint32_t)abs(e3); a5[4] += (uint32_t)abs(e4);
                }
#pragma unroll
                for (int kk = 0; kk < 5; kk++) tot[c][kk] = wave_sum64(a5[kk]);
            }
        }
    }

    // ---- per-candidate baseline: verbatim / constant, fixed order guess
    uint32_t best[NC], guess[NC];
    uint32_t d_type[NC], d_order[NC], d_prec[NC], d_porder[NC], d_method[NC], d_k[NC];
    int d_shift[NC];
    uint32_t fixed_mask = 0, lpc_mask = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t w = wst[c], sb = sbp[c];
        const u64 vb = (u64)8 + w + (u64)n * sb;
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
            const int32_t x0 = fcv_rt<MS>((uint32_t)c, k.pl[0], (NCH == 2) ? k.pr[0] : 0, w);
            uint32_t ne = 0;
            for (uint32_t i = lane; i < n; i += 64) ne |= (fcv_rt<MS>((uint32_t)c, k.pl[i], (NCH == 2) ? k.pr[i] : 0, w) != x0);
            constant = !__any(ne != 0);
        }
        if (mydbg && lane == 0) {
            for (int kk = 0; kk < 5; kk++) mydbg->cand[c].fixed_tot[kk] = tot[c][kk];
            mydbg->cand[c].fixed_guess = g;
        }
        if (constant) {
            const uint32_t cb = 8 + w + sb;
            if (cb < best[c]) { best[c] = cb; d_type[c] = 0; }
        }
        else {
            if (!(rbg >= (float)sb)) fixed_mask |= 1u << c;
            if (P.max_lpc_order > 0) lpc_mask |= 1u << c;
        }
    }
    FG_STAMP(2);

    // ================================================================ autocorrelation vectors (order-preserving fp64 chains)
    uint32_t nv = 0;
    const uint32_t mo = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
    if (lpc_mask && MAXO > 0 && mo > 0) {
        const uint32_t DSTR = FG_DH + FG_DK;
        const uint32_t cl = lane >> 4, l = lane & 15;
        const bool on = cl < (uint32_t)NC && l <= mo;
        const LDS double *cur = k.dbuf + (on ? cl : 0) * DSTR + FG_DH;
        const LDS double *hist = cur - (on ? l : 0);
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
                for (uint32_t j = lane; j < NC * FG_DH; j += 64) k.dbuf[(j / FG_DH) * DSTR + (j % FG_DH)] = 0.0;
                wave_lds_fence();
                for (uint32_t k0 = 0; k0 < vec_len; k0 += FG_DK) {
                    const uint32_t kn = (vec_len - k0) < FG_DK ? (vec_len - k0) : FG_DK;
                    for (uint32_t j = lane; j < kn; j += 64) {
                        const uint32_t i = k0 + j;
                        float wv;
                        uint32_t si;
                        bool zero = false;
                        if (part == 0) { wv = k.window[i]; si = i; }
                        else if (i < part) { wv = k.window[i]; si = sh + i; }
                        else if (i < 2 * part) { wv = k.window[n - 2 * part + i]; si = sh + i; }
                        else { wv = 0.0f; si = 0; zero = true; }
                        const int32_t L = k.pl[si], R = (NCH == 2) ? k.pr[si] : 0;
#pragma unroll
                        for (int c = 0; c < NC; c++) {
                            const int32_t x = fcv_rt<MS>((uint32_t)c, L, R, 0);
                            const float dd = zero ? 0.0f : (float)x * wv;
                            k.dbuf[c * DSTR + FG_DH + j] = (double)dd;
                        }
                    }
                    wave_lds_fence();
                    if (on) {
                        uint32_t j = 0;
                        for (; j + 8 <= kn; j += 8) {
                            double a[8], b[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) { a[u] = cur[j + u]; b[u] = hist[j + u]; }
#pragma unroll
                            for (int u = 0; u < 8; u++) acc = __builtin_fma(a[u], b[u], acc);
                        }
                        for (; j < kn; j++) acc = __builtin_fma(cur[j], hist[j], acc);
                    }
                    wave_lds_fence();
                    if (k0 + kn < vec_len) {
                        double t[(NC * FG_DH + 63) / 64];
#pragma unroll
                        for (int u = 0; u < (NC * FG_DH + 63) / 64; u++) {
                            const uint32_t j = u * 64 + lane;
                            t[u] = (j < NC * FG_DH) ? k.dbuf[(j / FG_DH) * DSTR + FG_DK + (j % FG_DH)] : 0.0;
                        }
                        wave_lds_fence();
#pragma unroll
                        for (int u = 0; u < (NC * FG_DH + 63) / 64; u++) {
                            const uint32_t j = u * 64 + lane;
                            if (j < NC * FG_DH) k.dbuf[(j / FG_DH) * DSTR + (j % FG_DH)] = t[u];
                        }
                        wave_lds_fence();
                    }
                }
                if (on) k.autoc[(cl * P.nvec + nv) * (MAXO + 1) + l] = acc;
                wave_lds_fence();
            }
            else if (punch) {
                // root - previous partial for lags < mo; lag mo keeps the partial (upstream quirk)
                const uint32_t total = (uint32_t)NC * (mo + 1);
                for (uint32_t j = lane; j < total; j += 64) {
                    const uint32_t c = j / (mo + 1), ll = j % (mo + 1);
                    LDS double *base = k.autoc + c * P.nvec * (MAXO + 1);
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
                mydbg->cand[c].autoc[v][ll] = k.autoc[(c * P.nvec + v) * (MAXO + 1) + ll];
            }
            if (lane < NC) mydbg->cand[lane].nvec = nv;
        }
    }
    FG_STAMP(4);

    // ================================================================ Levinson-Durbin, order guess, quantiser
    // lane = candidate * nvec + vector.  lres[idx] = order | prec<<8 | (shift&255)<<16 | ok<<24 | ran<<25
    if (nv > 0) {
        const uint32_t nidx = (uint32_t)NC * P.nvec;
        const uint32_t LS = nidx;
        LDS double *lpcw = k.dbuf;
        LDS float *lpf = (LDS float *)(k.dbuf + (size_t)mo * LS);
        const uint32_t idx = lane;
        if (idx < nidx) {
            const uint32_t c = idx / P.nvec, v = idx % P.nvec;
            const LDS double *A = k.autoc + (c * P.nvec + v) * (MAXO + 1);
            bool on = v < nv && ((lpc_mask >> c) & 1);
            if (on && A[0] == 0.0) on = false;
            uint32_t sb = sbp[0];
#pragma unroll
            for (int cc = 1; cc < NC; cc++) if (c == (uint32_t)cc) sb = sbp[cc];
            const double a0 = on ? A[0] : 1.0;
            const uint32_t overhead = sb + P.qlp_precision;
            const double scale = 0.5 / (double)n;
            double er = a0, bestb = 4294967295.0;
            uint32_t besti = 0;
            bool stopped = false;
            for (uint32_t i = 0; i < mo; i++) {
                double r = on ? -A[i + 1] : 0.0;
                for (uint32_t j = 0; j < i; j++) r -= lpcw[j * LS + idx] * (on ? A[i - j] : 0.0);
                r /= er;
                lpcw[i * LS + idx] = r;
                uint32_t j;
                for (j = 0; j < (i >> 1); j++) {
                    const double tmp = lpcw[j * LS + idx], t2 = lpcw[(i - 1 - j) * LS + idx];
                    lpcw[j * LS + idx] = tmp + r * t2;
                    lpcw[(i - 1 - j) * LS + idx] = t2 + r * tmp;
                }
                if (i & 1) { const double t = lpcw[j * LS + idx]; lpcw[j * LS + idx] = t + t * r; }
                er *= (1.0 - r * r);
                if (!stopped) {
                    const uint32_t o = i + 1;
                    const double bits = f_ebps(er, scale) * (double)(n - o) + (double)(o * overhead);
                    if (bits < bestb) { besti = i; bestb = bits; }
                    if (er == 0.0) stopped = true;
                }
            }
            const uint32_t ostar = besti + 1;
            double err2 = a0;
            for (uint32_t i = 0; i < ostar; i++) {
                double r = on ? -A[i + 1] : 0.0;
                for (uint32_t j = 0; j < i; j++) r -= lpcw[j * LS + idx] * (on ? A[i - j] : 0.0);
                r /= err2;
                lpcw[i * LS + idx] = r;
                uint32_t j;
                for (j = 0; j < (i >> 1); j++) {
                    const double tmp = lpcw[j * LS + idx], t2 = lpcw[(i - 1 - j) * LS + idx];
                    lpcw[j * LS + idx] = tmp + r * t2;
                    lpcw[(i - 1 - j) * LS + idx] = t2 + r * tmp;
                }
                if (i & 1) { const double t = lpcw[j * LS + idx]; lpcw[j * LS + idx] = t + t * r; }
                err2 *= (1.0 - r * r);
            }
            for (uint32_t jj = 0; jj < ostar; jj++) lpf[jj * LS + idx] = (float)(-lpcw[jj * LS + idx]);
            uint32_t result = 0;
            for (uint32_t j = 0; j < (uint32_t)MAXO; j++) k.qres[idx * MAXO + j] = 0;
            if (on) {
                bool ok = !(f_ebps(err2, 0.5 / (double)(n - ostar)) >= (double)sb);
                uint32_t prec = P.qlp_precision;
                if (sb <= 17) { const uint32_t lim = 32 - sb - ilog2_32(ostar); if (lim < prec) prec = lim; }
                int shift = 0;
                if (ok) {
                    const int p1 = (int)prec - 1;
                    const int32_t qmax = (1 << p1) - 1, qmin = -(1 << p1);
                    double cmax = 0.0;
                    for (uint32_t j = 0; j < ostar; j++) { const double dd = fabs((double)lpf[j * LS + idx]); if (dd > cmax) cmax = dd; }
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
                        for (uint32_t j = 0; j < ostar; j++) {
                            const double lpv = (double)lpf[j * LS + idx];
                            error += neg ? lpv / mul : lpv * mul;
                            const double rq = round(error);
                            int32_t qv = (int32_t)(i64)rq;
                            if (qv > qmax) qv = qmax; else if (qv < qmin) qv = qmin;
                            error -= (double)qv;
                            k.qres[idx * MAXO + j] = qv;
                        }
                        if (neg) shift = 0;
                    }
                }
                result = ostar | (prec << 8) | (((uint32_t)shift & 0xFF) << 16) | ((ok ? 1u : 0u) << 24) | (1u << 25);
            }
            k.lres[idx] = result;
        }
        wave_lds_fence();
    }
    FG_STAMP(5);

    // ================================================================ evaluation of the predictors: pass 0 = fixed, then
    // one pass per autocorrelation vector.  Each pass: FIR residual of every enabled candidate, |r| partition sums,
    // Rice parameter / partition order search, strict-< update of the best (libFLAC's candidate order).
    const uint32_t psz0 = n >> pmax0, ipp0 = psz0 >> 6, parts0 = 1u << pmax0;
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
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const uint32_t idx = (uint32_t)c * P.nvec + v;
                const uint32_t r = rfl(k.lres[idx]);
                order[c] = r & 0xFF; prec[c] = (r >> 8) & 0xFF; shift[c] = (int)(int8_t)((r >> 16) & 0xFF);
                if (((lpc_mask >> c) & 1) && ((r >> 24) & 1)) emask |= 1u << c;
                if (order[c] == 0) order[c] = 1;
#pragma unroll
                for (int j = 0; j < MAXO; j++) q[c][j] = (int32_t)rfl((uint32_t)k.qres[idx * MAXO + j]);
                if (mydbg && lane == 0) mydbg->cand[c].lpc_guess[v] = ((r >> 25) & 1) ? (r & 0xFF) : 0;
            }
            if (!emask) continue;
        }
        // ---- residual partition sums, lane p = partition p
        sum_t psum[NC];
        uint32_t ovf[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) { psum[c] = 0; ovf[c] = 0; }
        {
            uint32_t t = 0;
            for (uint32_t p = 0; p < parts0; p++) {
                sum_t a[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) a[c] = 0;
                for (uint32_t kk = 0; kk < ipp0; kk++, t++) {
                    const int i = (int)((t << 6) + lane);
                    int32_t l[MAXO + 1], r[MAXO + 1];
#pragma unroll
                    for (int j = 0; j <= MAXO; j++) { l[j] = k.pl[i - j]; r[j] = (NCH == 2) ? k.pr[i - j] : 0; }
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        if (!((emask >> c) & 1)) continue;
                        int32_t res;
                        const int32_t x0 = fcv_rt<MS>((uint32_t)c, l[0], r[0], 0);
                        if (!ACC64) {
                            int32_t s = 0;
#pragma unroll
                            for (int j = 0; j < MAXO; j++) s += __mul24(q[c][j], fcv_rt<MS>((uint32_t)c, l[j + 1], r[j + 1], 0));
                            res = x0 - (s >> shift[c]);
                        }
                        else {
                            i64 s = 0;
#pragma unroll
                            for (int j = 0; j < MAXO; j++) s += (i64)q[c][j] * (i64)fcv_rt<MS>((uint32_t)c, l[j + 1], r[j + 1], 0);
                            const i64 rr = (i64)x0 - (s >> shift[c]);
                            if (rr <= (i64)INT32_MIN || rr > (i64)INT32_MAX) ovf[c] = 1;
                            res = (int32_t)rr;
                        }
                        if ((uint32_t)i >= order[c]) a[c] += (uint32_t)abs(res);
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (!((emask >> c) & 1)) continue;
                    sum_t tt;
                    if (ACC64) tt = (sum_t)wave_sum64((u64)a[c]);
                    else tt = (sum_t)wave_sum((uint32_t)a[c]);
                    if ((uint32_t)lane == p) psum[c] = tt;
                }
            }
        }
        // ---- Rice search per candidate
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (!((emask >> c) & 1)) continue;
            uint32_t est = 0;
            if (!(ACC64 && __any(ovf[c] != 0))) {
                const uint32_t limit = P.rice_limit, sb = sbp[c];
                const bool wrap32 = (sb + 4) < (32 - ilog2_32(psz0));
                u64 s = (u64)psum[c];
                if (wrap32) s &= 0xFFFFFFFFull;
                uint32_t best_bits = 0, bpo = 0, kb = 0;
                for (int po = (int)pmax0; po >= (int)pmin0; po--) {
                    const uint32_t parts = 1u << po;
                    const uint32_t pbase = n >> po;
                    uint32_t np = pbase, dv = 0x40000u / pbase;
                    if (lane == 0) { np -= order[c]; dv = 0x40000u / np; }
                    uint32_t kr = 0;
                    if (s >= 2) {
                        const u64 qv = ((s - 1) * dv) >> 18;
                        if (qv != 0) kr = ilog2_64(qv) + 1;
                    }
                    if (kr >= limit) kr = limit - 1;
                    u64 pb = (u64)4 + (u64)(1 + kr) * np + (kr ? (s >> (kr - 1)) : (s << 1)) - (np >> 1);
                    if (pb > 0xFFFFFFFFull) pb = 0xFFFFFFFFull;
                    if ((uint32_t)lane >= parts) pb = 0;
                    u64 total;
                    if (__any(pb >> 25)) total = wave_sum64(pb) + 6;
                    else total = (u64)wave_sum((uint32_t)pb) + 6;
                    const uint32_t bits = total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total;
                    if (best_bits == 0 || bits < best_bits) { best_bits = bits; bpo = (uint32_t)po; kb = kr; }
                    if (po > (int)pmin0) {
                        // merge pairs: lane p <- s[2p] + s[2p+1]
                        const uint32_t lo = (uint32_t)s, hi = (uint32_t)(s >> 32);
                        const int src = (lane * 2) & 63;
                        u64 s0 = (uint32_t)__shfl((int)lo, src), s1 = (uint32_t)__shfl((int)lo, src + 1);
                        if (__any(hi != 0)) {
                            s0 |= (u64)(uint32_t)__shfl((int)hi, src) << 32;
                            s1 |= (u64)(uint32_t)__shfl((int)hi, src + 1) << 32;
                        }
                        s = ((uint32_t)lane < (parts >> 1)) ? s0 + s1 : 0;
                    }
                }
                est = kind == 0 ? (8 + wst[c] + order[c] * sb) : (8 + wst[c] + 4 + 5 + order[c] * (prec[c] + sb));
                if (best_bits < 0xFFFFFFFFu - est) est += best_bits; else est = 0xFFFFFFFFu;
                if (est > 0 && est < best[c]) {
                    best[c] = est;
                    d_type[c] = kind == 0 ? 2 : 3; d_order[c] = order[c]; d_prec[c] = prec[c]; d_shift[c] = shift[c];
                    d_porder[c] = bpo; d_k[c] = kb;
                    d_method[c] = __any(((uint32_t)lane < (1u << bpo)) && kb >= 15) ? 1 : 0;
                    if (kind == 1 && lane < MAXO) k.bestq[c * MAXO + lane] = k.qres[((uint32_t)c * P.nvec + (pass - 1)) * MAXO + lane];
                }
            }
            if (mydbg && lane == 0) {
                if (kind == 0) mydbg->cand[c].fixed_bits = est;
                else mydbg->cand[c].lpc_bits[pass - 1] = est;
            }
        }
        wave_lds_fence();
    }
    FG_STAMP(6);

    // ================================================================ channel assignment
    uint32_t ca = 0, sub0 = 0, sub1 = 1;
    if (MS) {
        if (d.forced_ca != 0xFF) ca = d.forced_ca;
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
                for (uint32_t j = 0; j < FG_MAX_ORDER; j++) dc->qlp[j] = (d_type[c] == 3 && j < d_order[c] && j < (uint32_t)MAXO) ? k.bestq[c * MAXO + j] : 0;
            }
            if (d_type[c] >= 2 && (uint32_t)lane < (1u << d_porder[c])) mydbg->cand[c].rice_params[lane] = d_k[c];
        }
    }
    FG_STAMP(7);

    // ================================================================ pack: header, subframes, padding, CRC-16
    FastBW bw;
    bw.outw = (uint32_t *)(out + (size_t)d.out_slot * P.slot_bytes);
    bw.slot_words = P.slot_bytes / 4; bw.bitpos = 0; bw.wbase = 0; bw.err = 0;
    for (uint32_t j = lane; j < FG_WINW + 2; j += 64) k.win[j] = 0;
    wave_lds_fence();
    {   // frame header: assembled by lane 0 in LDS, emitted one byte per lane
        LDS uint8_t *hb = (LDS uint8_t *)k.misc;
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
        bw_round(bw, k.win, lane, 0, 0, v, b, b);
    }
    for (uint32_t si = 0; si < (uint32_t)NCH; si++) {
        const uint32_t c = MS ? (si == 0 ? sub0 : sub1) : si;
        // select the decision of candidate c (wave-uniform)
        uint32_t type = d_type[0], order = d_order[0], w = wst[0], sb = sbp[0], prec = d_prec[0], po = d_porder[0], method = d_method[0],
                 kv = d_k[0];
        int shift = d_shift[0];
#pragma unroll
        for (int cc = 1; cc < NC; cc++)
            if (c == (uint32_t)cc) {
                type = d_type[cc]; order = d_order[cc]; w = wst[cc]; sb = sbp[cc]; prec = d_prec[cc]; po = d_porder[cc]; method = d_method[cc];
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
        // ---- one round for everything in front of the residual: lane 0 = subframe header byte (+ wasted-bits unary),
        // lanes 1..order = warm-up samples, then precision/shift, coefficients, coding method + partition order
        {
            uint32_t pv = 0, pb = 0, val = 0, vb = 0, nb = 0;
            const bool pred = type >= 2;
            const uint32_t nw = type == 0 ? 1 : (pred ? order : 0);   // sample fields in this round
            if (lane == 0) { pv = hdr | (w ? 1 : 0); pb = 8; if (w) { val = 1; vb = 1; nb = w; } }
            else if ((uint32_t)lane <= nw) {
                val = (uint32_t)fcv_rt<MS>(c, k.pl[lane - 1], (NCH == 2) ? k.pr[lane - 1] : 0, w) & mask; vb = sb; nb = sb;
            }
            else if (type == 3 && (uint32_t)lane == order + 1) { pv = prec - 1; pb = 4; val = (uint32_t)shift & 31; vb = 5; nb = 5; }
            else if (type == 3 && (uint32_t)lane <= 2 * order + 1) {
                val = (uint32_t)k.bestq[c * MAXO + (lane - order - 2)] & ((1u << prec) - 1); vb = prec; nb = prec;
            }
            else if (pred && (uint32_t)lane == (type == 3 ? 2 * order + 2 : order + 1)) { val = (method << 4) | po; vb = 6; nb = 6; }
            bw_round(bw, k.win, lane, pv, pb, val, vb, nb);
        }
        if (type == 1) {
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                bw_round(bw, k.win, lane, 0, 0, (uint32_t)fcv_rt<MS>(c, k.pl[i], (NCH == 2) ? k.pr[i] : 0, w) & mask, sb, sb);
            }
        }
        else if (type >= 2) {
            int32_t q[MAXO];
            if (type == 3) {
#pragma unroll
                for (int j = 0; j < MAXO; j++) q[j] = (int32_t)rfl((uint32_t)k.bestq[c * MAXO + j]);
            }
            else {
                const uint32_t g = order;
                const int32_t c0 = g == 0 ? 0 : (int32_t)g, c1 = g < 2 ? 0 : (g == 2 ? -1 : g == 3 ? -3 : -6);
                const int32_t c2 = g < 3 ? 0 : (g == 3 ? 1 : 4), c3 = g < 4 ? 0 : -1;
#pragma unroll
                for (int j = 0; j < MAXO; j++) q[j] = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : j == 3 ? c3 : 0;
                shift = 0;
            }
            const uint32_t plen = method ? 5 : 4;
            const uint32_t psz = n >> po, ipp = psz >> 6;
            uint32_t t = 0;
            for (uint32_t p = 0; p < (1u << po); p++) {
                const uint32_t kr = rl(kv, (int)p);
                for (uint32_t kk = 0; kk < ipp; kk++, t++) {
                    const int i = (int)((t << 6) + lane);
                    int32_t xw[MAXO + 1];
#pragma unroll
                    for (int j = 0; j <= MAXO; j++) xw[j] = fcv_rt<MS>(c, k.pl[i - j], (NCH == 2) ? k.pr[i - j] : 0, 0);
                    int32_t r;
                    if (!ACC64) {
                        int32_t s = 0;
#pragma unroll
                        for (int j = 0; j < MAXO; j++) s += __mul24(q[j], xw[j + 1]);
                        r = xw[0] - (s >> shift);
                    }
                    else {
                        i64 s = 0;
#pragma unroll
                        for (int j = 0; j < MAXO; j++) s += (i64)q[j] * (i64)xw[j + 1];
                        r = (int32_t)((i64)xw[0] - (s >> shift));
                    }
                    uint32_t pv = 0, pb = 0, val = 0, vb = 0, nb = 0;
                    if ((uint32_t)i >= order) {
                        const uint32_t u = ((uint32_t)r << 1) ^ (uint32_t)(r >> 31);
                        val = (1u << kr) | (u & ((1u << kr) - 1));
                        vb = kr + 1;
                        nb = (u >> kr) + 1 + kr;
                        if (kk == 0 && (uint32_t)i == (p == 0 ? order : p * psz)) { pv = kr; pb = plen; }
                    }
                    bw_round(bw, k.win, lane, pv, pb, val, vb, nb);
                }
            }
        }
    }
    FG_STAMP(8);
    // ---- zero-pad to a byte, CRC-16 over the whole frame (64 lanes over interleaved words), append
    if (bw.bitpos & 7) bw.bitpos += 8 - (bw.bitpos & 7);
    bw_flush(bw, k.win, lane, bw.bitpos);
    bw_flush_all(bw, k.win, lane);
    __threadfence_block();
    {
        const uint32_t nbytes = bw.bitpos >> 3;
        const uint32_t W = nbytes >> 2, tail = nbytes & 3;
        const uint32_t pad = (64 - (W & 63)) & 63, T = (W + pad) >> 6;
        uint32_t s = 0;
        const LDS uint16_t *t0 = k.crct, *thi = k.crct + 256, *tlo = k.crct + 512;
        for (uint32_t t = 0; t < T; t++) {
            const int qi = (int)(t * 64 + lane) - (int)pad;
            uint32_t wv = 0;
            if (qi >= 0) wv = __builtin_bswap32(__builtin_nontemporal_load(&bw.outw[qi]));
            s = thi[s >> 8] ^ tlo[s & 0xFF];
            uint32_t cw = 0;
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 24)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 16)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ (wv >> 8)) & 0xFF];
            cw = ((cw << 8) & 0xFFFF) ^ t0[((cw >> 8) ^ wv) & 0xFF];
            s ^= cw;
        }
        s = gf16_mul(s, k.misc[64 + (63 - lane)]);
        uint32_t crc = wave_xor32(s);
        if (tail) {
            const uint32_t wv = W < bw.slot_words ? __builtin_bswap32(__builtin_nontemporal_load(&bw.outw[W])) : 0;
            for (uint32_t b = 0; b < tail; b++) crc = ((crc << 8) & 0xFFFF) ^ t0[((crc >> 8) ^ (wv >> (24 - 8 * b))) & 0xFF];
        }
        bw_put(bw, k.win, lane, crc, 16);
        bw_flush_all(bw, k.win, lane);
    }
    FG_STAMP(9);
    if (lane == 0) {
        FgBlockResult *r = &results[d.out_slot];
        r->bytes = bw.bitpos >> 3; r->ca = ca; r->err = err | bw.err; r->reserved = 1;
#pragma unroll
        for (int c = 0; c < 4; c++) r->best_bits[c] = c < NC ? best[c < NC ? c : 0] : 0;
    }
#undef FG_STAMP
}

}  // namespace
