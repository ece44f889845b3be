// flac_enc_wave_impl.h -- the specialised FLAC frame encoder with one wavefront per predictor candidate.
//
// Same lane = segment mapping and the same stages as flac_enc_fast_impl.h (which this file builds on), but a block is a
// workgroup of NC wavefronts (L, R, M, S for mid-side stereo; L, R; or one): wave c stages a share of the samples, then
// runs the whole analysis of candidate c on its own -- error sums, autocorrelation, Levinson-Durbin, FIR evaluation, Rice
// search -- with no cross-wave traffic until the decisions meet in LDS.  Waves 0 .. NCH-1 then pack one subframe each:
// both measure their subframe (pass A), one barrier gives wave 1 its start position, both emit into their own LDS window,
// and wave 1 takes over the last partial word and the CRC lane states of wave 0 to finish the frame.
// Four waves per block share the staged samples, so a CU holds four times the wavefronts for the same LDS, and the
// per-block latency drops to that of one candidate.
#pragma once
#define FG_CRCT __attribute__((address_space(3)))
#include "flac_enc_fast_impl.h"

namespace {

struct WaveDecision {        // what a wave publishes about its candidate
    uint32_t best, type, order, prec, porder, method, sbps;
    int32_t shift;
};

template <bool MS, int NCH, int MAXO, bool ACC64>
__global__ void __launch_bounds__(64 * (MS ? 4 : NCH))
fg_encode_wave_kernel(const void *pcm, const FgBlockDesc *descs, const float *windows, FgEncParams P, uint8_t *out,
                      FgBlockResult *results, FgDebugRec *dbg, const uint16_t *crctab)
{
    constexpr int NC = MS ? 4 : NCH;
    constexpr int NT = 64 * NC;
    typedef typename FastTypes<ACC64>::sum_t sum_t;
    typedef typename FastTypes<ACC64>::samp_t samp_t;
    constexpr uint32_t PADE = FastTypes<ACC64>::PADE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FgBlockDesc d = descs[blockIdx.x];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t c = rfl((uint32_t)tid >> 6);     // this wave's candidate
    const uint32_t n = d.n;
    const uint32_t seg = n >> 6;                    // samples per lane
    const uint32_t rstr = seg + PADE;               // LDS row stride (elements)
    // ---- LDS carve
    LDS unsigned char *lbase = (LDS unsigned char *)smem;
    uint32_t off = 0;
#define FG_CARVE(type, bytes) (LDS type *)(lbase + off); off += (uint32_t)(((bytes) + 15) & ~15u)
    LDS samp_t *sL = FG_CARVE(samp_t, (P.sig_stride + 256) * sizeof(samp_t));      // 64 rows, up to 4 elements of skew each
    LDS samp_t *sR = FG_CARVE(samp_t, NCH == 2 ? (P.sig_stride + 256) * sizeof(samp_t) : 16);
    // per-wave analysis scratch (autocorrelation staging row, Levinson-Durbin work space); the frame-bit windows of the
    // packing waves live in the same region later
    const uint32_t lev = (P.nvec * (P.max_lpc_order ? P.max_lpc_order : 1) * 12 + 64 + 15) & ~15u;
    const uint32_t wbytes = lev > FGS_DSTR * 8 ? lev : FGS_DSTR * 8;
    uint32_t ubytes = NC * wbytes;
    if (ubytes < NCH * (FGS_FBW + 2) * 4) ubytes = NCH * (FGS_FBW + 2) * 4;
    LDS unsigned char *ureg = FG_CARVE(unsigned char, ubytes);
    LDS double *wscr = (LDS double *)(ureg + c * wbytes);
    LDS double *autoc = FG_CARVE(double, NC * P.nvec * (MAXO + 1) * 8);
    LDS int32_t *qres = FG_CARVE(int32_t, NC * P.nvec * MAXO * 4);
    LDS uint32_t *lres = FG_CARVE(uint32_t, NC * P.nvec * 4);
    LDS int32_t *bestq = FG_CARVE(int32_t, NC * MAXO * 4);
    LDS uint32_t *dk = FG_CARVE(uint32_t, NC * 64 * 4);                 // Rice parameters of the best predictor, per lane
    LDS WaveDecision *dec = FG_CARVE(WaveDecision, NC * sizeof(WaveDecision));
    LDS uint16_t *crct = FG_CARVE(uint16_t, 1536 * 2);
    LDS uint32_t *mult = FG_CARVE(uint32_t, 64 * 4);                    // x^(32 j) mod P, j = 0..63
    LDS uint32_t *scrw = FG_CARVE(uint32_t, NT * 4);                    // one scratch word per thread (parked stores)
    LDS uint32_t *hand = FG_CARVE(uint32_t, 72 * 4);                    // wave 0 -> wave 1: CRC lane states, tail word, positions
    LDS uint32_t *hdrb = FG_CARVE(uint32_t, 32);                        // frame header bytes
    LDS uint32_t *flags = FG_CARVE(uint32_t, 16);
#undef FG_CARVE
    const float *window = windows + d.win_off;
    for (int j = tid; j < 768; j += NT) { crct[j] = crctab[j]; crct[768 + j] = crctab[1024 + j]; }
    if (tid < 64) mult[tid] = crctab[768 + tid];
    if (tid == 0) { flags[0] = 0; flags[1] = 0; }
    FgDebugRec *mydbg = dbg ? dbg + d.out_slot : nullptr;
#define FG_STAMP(i) do { if (mydbg && tid == 0) mydbg->t[i] = clock64(); } while (0)
    FG_STAMP(0);
    uint32_t err = 0;
    const LDS samp_t *rowL = sL + (uint32_t)lane * rstr, *rowR = sR + (uint32_t)lane * rstr;
    const uint32_t magic = 0xFFFFFFFFu / seg + 1;
#define FG_SADDR(g) ((g) + __umulhi((g), magic) * PADE)
    // candidate value of this wave from the two channel samples (c is wave-uniform)
    auto cand = [&](int32_t l, int32_t r) __attribute__((always_inline)) -> int32_t {
        if (!MS) return c == 0 ? l : r;
        return c == 0 ? l : c == 1 ? r : c == 2 ? ((l + r) >> 1) : (l - r);
    };

    // ================================================================ stage: HBM -> LDS, all waves
    {
        const int32_t lim = (int32_t)(P.bps - 1);
        uint32_t bad = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += 4 * NT) {
            int32_t a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * NT + tid;
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
                const uint32_t i = i0 + u * NT + tid;
                if (i < n) {
                    if (P.bps < 32) bad |= (uint32_t)(((a[u] ^ (a[u] >> 31)) >> lim) | ((b[u] ^ (b[u] >> 31)) >> lim));
                    const uint32_t ad = FG_SADDR(i);
                    sL[ad] = (samp_t)a[u];
                    if (NCH == 2) sR[ad] = (samp_t)b[u];
                }
            }
        }
        if (__any(bad != 0)) err |= FG_ERR_RANGE;
    }
    __syncthreads();
    if (err && lane == 0) flags[1] = err;          // read by the finishing wave, several barriers later
    FG_STAMP(1);

    uint32_t pmax0 = 0;
    { uint32_t b = n; while (!(b & 1)) { pmax0++; b >>= 1; } if (pmax0 > 15) pmax0 = 15; }
    if (P.max_po < pmax0) pmax0 = P.max_po;
    const uint32_t pmin0 = P.min_po < pmax0 ? P.min_po : pmax0;

    // ================================================================ wasted bits + fixed-predictor error sums of candidate c
    uint32_t wst, sbp;
    u64 tot[5];
    {
        sum_t acc[5] = {0, 0, 0, 0, 0};
        uint32_t orv = 0;
        int32_t p1 = 0, q1 = 0, q2 = 0, q3 = 0;       // previous value and previous 1st..3rd differences
#pragma unroll 1
        for (int s = -4; s < (int)seg; s++) {
            int32_t l = 0, r = 0;
            if (s >= 0) { l = rowL[s]; r = (NCH == 2) ? rowR[s] : 0; }
            else if (lane > 0) { l = rowL[(int)seg + s - (int)rstr]; r = (NCH == 2) ? rowR[(int)seg + s - (int)rstr] : 0; }
            // the sums run over samples 4 .. n-1 (libFLAC hands fixed.c the signal shifted by the maximum fixed order)
            const bool on = s >= 0 && (lane > 0 || s >= 4);
            const int32_t v = cand(l, r);
            const int32_t e1 = v - p1, e2 = e1 - q1, e3 = e2 - q2, e4 = e3 - q3;
            p1 = v; q1 = e1; q2 = e2; q3 = e3;
            if (s >= 0) orv |= (uint32_t)v;
            if (on) { acc[0] += fabs32(v); acc[1] += fabs32(e1); acc[2] += fabs32(e2); acc[3] += fabs32(e3); acc[4] += fabs32(e4); }
        }
        const uint32_t o = wave_or32(orv);
        uint32_t w = o ? (uint32_t)__builtin_ctz(o) : 0;
        const uint32_t nominal = P.bps + ((MS && c == 3) ? 1u : 0u);
        if (w > nominal) w = nominal;
        wst = w; sbp = nominal - w;
#pragma unroll
        for (int kk = 0; kk < 5; kk++) tot[kk] = ACC64 ? wave_sum64((u64)acc[kk]) : (u64)wave_sum((uint32_t)acc[kk]);
        // Blocks in which some candidate has wasted bits (all samples share trailing zero bits: rare) are handed to the
        // generic kernel; the waves keep going (their results are dropped) so that every barrier is met by all of them.
        if (wst && lane == 0) flags[0] = 1;
    }

    // ---- baseline of candidate c: verbatim / constant, fixed order guess
    uint32_t best, guess;
    uint32_t d_type = 1, d_order = 0, d_prec = 0, d_porder = 0, d_method = 0, d_k = 0;
    int d_shift = 0;
    bool fixed_on = false, lpc_on = false;
    {
        const uint32_t sb = sbp;
        const u64 vb = (u64)8 + (u64)n * sb;
        best = vb < 0xFFFFFFFFull ? (uint32_t)vb : 0xFFFFFFFFu;
        const u64 m34 = tot[3] < tot[4] ? tot[3] : tot[4];
        const u64 m234 = tot[2] < m34 ? tot[2] : m34;
        const u64 m1234 = tot[1] < m234 ? tot[1] : m234;
        uint32_t g;
        u64 tg;
        if (tot[0] <= m1234) { g = 0; tg = tot[0]; }
        else if (tot[1] <= m234) { g = 1; tg = tot[1]; }
        else if (tot[2] <= m34) { g = 2; tg = tot[2]; }
        else if (tot[3] <= tot[4]) { g = 3; tg = tot[3]; }
        else { g = 4; tg = tot[4]; }
        guess = g;
        const double len = (double)(n - 4);
        const float rbg = (float)((tg > 0) ? log(FG_LN2 * (double)tg / len) / FG_LN2 : 0.0);
        bool constant = false;
        if (tot[1] == 0) {
            const int32_t x0 = cand(sL[0], (NCH == 2) ? sR[0] : 0);
            uint32_t ne = 0;
#pragma unroll 1
            for (uint32_t s = 0; s < seg; s++) ne |= (cand(rowL[s], (NCH == 2) ? rowR[s] : 0) != x0);
            constant = !__any(ne != 0);
        }
        if (mydbg && lane == 0) {
            for (int kk = 0; kk < 5; kk++) mydbg->cand[c].fixed_tot[kk] = tot[kk];
            mydbg->cand[c].fixed_guess = g;
        }
        if (constant) {
            const uint32_t cb = 8 + sb;
            if (cb < best) { best = cb; d_type = 0; }
        }
        else {
            if (!(rbg >= (float)sb)) fixed_on = true;
            if (P.max_lpc_order > 0) lpc_on = true;
        }
    }
    FG_STAMP(2);

    // ================================================================ autocorrelation vectors of candidate c
    // lane = lag.  Per chunk of FGS_DK samples the windowed signal is staged as doubles (FGS_DH history entries in front);
    // lane l reads d[j - l], the d[j] operand is lane 0's own value, broadcast inside the FMA (DPP row_newbcast:0).
    uint32_t nv = 0;
    const uint32_t mo = P.max_lpc_order >= n ? n - 1 : P.max_lpc_order;
    if (lpc_on && MAXO > 0 && mo > 0) {
        const uint32_t l = (uint32_t)lane;
        const bool on = l <= mo;
        LDS double *drow = wscr;
        const LDS double *hist = drow + FGS_DH - (on ? l : 0);
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
                if (lane < FGS_DH) drow[lane] = 0.0;
                wave_lds_fence();
                float wv[FGS_DK / 64];
                uint32_t si[FGS_DK / 64];
                auto fetch = [&](uint32_t k0) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < FGS_DK / 64; u++) {
                        const uint32_t i = k0 + u * 64 + lane;
                        float w = 0.0f;
                        uint32_t s_ = 0;
                        if (i < vec_len) {
                            if (part == 0) { w = window[i]; s_ = i; }
                            else if (i < part) { w = window[i]; s_ = sh + i; }
                            else if (i < 2 * part) { w = window[n - 2 * part + i]; s_ = sh + i; }
                        }
                        wv[u] = w; si[u] = s_;
                    }
                };
                fetch(0);
                for (uint32_t k0 = 0; k0 < vec_len; k0 += FGS_DK) {
                    const uint32_t kn = (vec_len - k0) < FGS_DK ? (vec_len - k0) : FGS_DK;
#pragma unroll
                    for (int u = 0; u < FGS_DK / 64; u++) {
                        const uint32_t j = u * 64 + lane;
                        if (j < kn) {
                            const uint32_t ad = FG_SADDR(si[u]);
                            const int32_t x = cand(sL[ad], (NCH == 2) ? sR[ad] : 0);
                            const bool zero = part != 0 && (k0 + j) >= 2 * part;
                            const float dd = zero ? 0.0f : (float)x * wv[u];
                            drow[FGS_DH + j] = (double)dd;
                        }
                    }
                    if (k0 + FGS_DK < vec_len) fetch(k0 + FGS_DK);
                    wave_lds_fence();
                    if (on) {
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
                        const double t = (lane < FGS_DH) ? drow[FGS_DK + lane] : 0.0;
                        wave_lds_fence();
                        if (lane < FGS_DH) drow[lane] = t;
                        wave_lds_fence();
                    }
                }
                if (on) autoc[(c * P.nvec + nv) * (MAXO + 1) + l] = acc;
                wave_lds_fence();
            }
            else if (punch) {
                // root - previous partial for lags < mo; lag mo keeps the partial (upstream quirk)
                if (l <= mo) {
                    LDS double *base = autoc + c * P.nvec * (MAXO + 1);
                    const double prev = base[(nv - 1) * (MAXO + 1) + l];
                    base[nv * (MAXO + 1) + l] = (l < mo) ? base[l] - prev : prev;
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
        if (mydbg) {
            for (uint32_t j = lane; j < nv * (mo + 1); j += 64) {
                const uint32_t v = j / (mo + 1), ll = j % (mo + 1);
                mydbg->cand[c].autoc[v][ll] = autoc[(c * P.nvec + v) * (MAXO + 1) + ll];
            }
            if (lane == 0) mydbg->cand[c].nvec = nv;
        }
    }
    FG_STAMP(4);

    // ================================================================ Levinson-Durbin, order guess, quantiser
    // lane = vector of this wave's candidate.  lres[idx] = order | prec<<8 | (shift&255)<<16 | ok<<24 | ran<<25
    if (nv > 0) {
        const uint32_t LS = P.nvec;
        LDS double *lpcw = wscr;
        LDS float *lpf = (LDS float *)(wscr + (size_t)mo * LS);
        const uint32_t v = (uint32_t)lane;
        if (v < P.nvec) {
            const uint32_t idx = c * P.nvec + v;
            const LDS double *A = autoc + idx * (MAXO + 1);
            bool on = v < nv && lpc_on;
            if (on && A[0] == 0.0) on = false;
            const uint32_t sb = sbp;
            const double a0 = on ? A[0] : 1.0;
            const uint32_t overhead = sb + P.qlp_precision;
            const double scale = 0.5 / (double)n;
            double er = a0, bestb = 4294967295.0;
            uint32_t besti = 0;
            bool stopped = false;
            for (uint32_t i = 0; i < mo; i++) {
                double r = on ? -A[i + 1] : 0.0;
                for (uint32_t j = 0; j < i; j++) r -= lpcw[j * LS + v] * (on ? A[i - j] : 0.0);
                r /= er;
                lpcw[i * LS + v] = r;
                uint32_t j;
                for (j = 0; j < (i >> 1); j++) {
                    const double tmp = lpcw[j * LS + v], t2 = lpcw[(i - 1 - j) * LS + v];
                    lpcw[j * LS + v] = tmp + r * t2;
                    lpcw[(i - 1 - j) * LS + v] = t2 + r * tmp;
                }
                if (i & 1) { const double t = lpcw[j * LS + v]; lpcw[j * LS + v] = t + t * r; }
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
                for (uint32_t j = 0; j < i; j++) r -= lpcw[j * LS + v] * (on ? A[i - j] : 0.0);
                r /= err2;
                lpcw[i * LS + v] = r;
                uint32_t j;
                for (j = 0; j < (i >> 1); j++) {
                    const double tmp = lpcw[j * LS + v], t2 = lpcw[(i - 1 - j) * LS + v];
                    lpcw[j * LS + v] = tmp + r * t2;
                    lpcw[(i - 1 - j) * LS + v] = t2 + r * tmp;
                }
                if (i & 1) { const double t = lpcw[j * LS + v]; lpcw[j * LS + v] = t + t * r; }
                err2 *= (1.0 - r * r);
            }
            for (uint32_t jj = 0; jj < ostar; jj++) lpf[jj * LS + v] = (float)(-lpcw[jj * LS + v]);
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
                    for (uint32_t j = 0; j < ostar; j++) { const double dd = fabs((double)lpf[j * LS + v]); if (dd > cmax) cmax = dd; }
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
                            const double lpv = (double)lpf[j * LS + v];
                            error += neg ? lpv / mul : lpv * mul;
                            const double rq = round(error);
                            int32_t qv = (int32_t)(i64)rq;
                            if (qv > qmax) qv = qmax; else if (qv < qmin) qv = qmin;
                            error -= (double)qv;
                            qres[idx * MAXO + j] = qv;
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

    // ================================================================ evaluation of candidate c: pass 0 = fixed predictor, then
    // one pass per autocorrelation vector (FIR over the lane's segment, Rice search, strict-< update of the best)
#pragma unroll 1
    for (uint32_t pass = 0; pass < 1 + nv; pass++) {
        uint32_t order, prec;
        int32_t q[MAXO];
        int shift;
        const int kind = pass == 0 ? 0 : 1;
        if (pass == 0) {
            if (!fixed_on) continue;
            const uint32_t g = guess;
            order = g; shift = 0; prec = 0;
            const int32_t c0 = g == 0 ? 0 : (int32_t)g, c1 = g < 2 ? 0 : (g == 2 ? -1 : g == 3 ? -3 : -6);
            const int32_t c2 = g < 3 ? 0 : (g == 3 ? 1 : 4), c3 = g < 4 ? 0 : -1;
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : j == 3 ? c3 : 0;
        }
        else {
            const uint32_t idx = c * P.nvec + (pass - 1);
            const uint32_t r = rfl(lres[idx]);
            order = r & 0xFF; prec = (r >> 8) & 0xFF; shift = (int)(int8_t)((r >> 16) & 0xFF);
            const bool en = lpc_on && ((r >> 24) & 1);
            if (order == 0) order = 1;
            // the coefficients: one LDS read for all of them, then broadcasts
            const int32_t qv = (lane < MAXO) ? qres[idx * MAXO + lane] : 0;
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = (int32_t)rl((uint32_t)qv, j);
            if (mydbg && lane == 0) mydbg->cand[c].lpc_guess[pass - 1] = ((r >> 25) & 1) ? (r & 0xFF) : 0;
            if (!en) continue;
        }
        // ---- FIR over the segment
        sum_t psum = 0;
        uint32_t ovf = 0;
        {
            int32_t h[MAXO];
#pragma unroll
            for (int j = 0; j < MAXO; j++) {
                int32_t x = 0;
                if (lane > 0) x = cand(rowL[(int)seg - 1 - j - (int)rstr], (NCH == 2) ? rowR[(int)seg - 1 - j - (int)rstr] : 0);
                h[(MAXO - 1 - j) % MAXO] = x;
            }
            auto step = [&](int u, uint32_t s, bool guard) __attribute__((always_inline)) {
                const int32_t x = cand(rowL[s], (NCH == 2) ? rowR[s] : 0);
                int32_t res;
                if (!ACC64) res = x - (fir24<MAXO>(q, h, u) >> shift);
                else {
                    const i64 rr = (i64)x - (fir64<MAXO>(q, h, u) >> shift);
                    if (rr <= (i64)INT32_MIN || rr > (i64)INT32_MAX) ovf = 1;
                    res = (int32_t)rr;
                }
                h[u] = x;
                if (!guard || lane > 0 || s >= order) psum += fabs32(res);
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
        }
        // ---- Rice parameter / partition order search (partition p of order po lives in lane p * (64 >> po))
        {
            u64 sv = (u64)psum;
            uint32_t best_bits = 0, bpo = 0, kb = 0;
            const bool dead = ACC64 && __any(ovf != 0);
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
            if ((sbp + 4) < (32 - ilog2_32(psz0))) sv &= 0xFFFFFFFFull;
            const uint32_t limit = P.rice_limit;
            for (int po = (int)pmax0; po >= (int)pmin0; po--) {
                const uint32_t stride = 64u >> po;
                const bool valid = ((uint32_t)lane & (stride - 1)) == 0;
                const uint32_t pbase = n >> po;
                auto div18 = [&](uint32_t x) __attribute__((always_inline)) -> uint32_t {
                    uint32_t qd = (uint32_t)(262144.0f * __builtin_amdgcn_rcpf((float)x));
                    const int32_t r = (int32_t)(0x40000u - qd * x);
                    if (r < 0) qd--;
                    else if ((uint32_t)r >= x) qd++;
                    return qd;
                };
                const uint32_t dv_all = div18(pbase), dv0 = div18(pbase - order);
                const u64 s = sv;
                uint32_t np = pbase, dv = dv_all;
                if (lane == 0) { np -= order; dv = dv0; }
                uint32_t kr = 0;
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
                const uint32_t bits = total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total;
                if (best_bits == 0 || bits < best_bits) { best_bits = bits; bpo = (uint32_t)po; kb = kr; }
                if (po > (int)pmin0) merge(6 - (uint32_t)po);
            }
            uint32_t est = 0;
            if (!dead) {
                const uint32_t sb = sbp;
                est = kind == 0 ? (8 + order * sb) : (8 + 4 + 5 + order * (prec + sb));
                if (best_bits < 0xFFFFFFFFu - est) est += best_bits; else est = 0xFFFFFFFFu;
                if (est > 0 && est < best) {
                    best = est;
                    d_type = kind == 0 ? 2 : 3; d_order = order; d_prec = prec; d_shift = shift;
                    d_porder = bpo; d_k = kb;
                    d_method = __any((((uint32_t)lane & ((64u >> bpo) - 1)) == 0) && kb >= 15) ? 1 : 0;
                    if (kind == 1 && lane < MAXO) bestq[c * MAXO + lane] = qres[(c * P.nvec + (pass - 1)) * MAXO + lane];
                }
            }
            if (mydbg && lane == 0) {
                if (kind == 0) mydbg->cand[c].fixed_bits = est;
                else mydbg->cand[c].lpc_bits[pass - 1] = est;
            }
        }
        wave_lds_fence();
    }
    // ---- publish the decision of candidate c
    if (lane == 0) {
        LDS WaveDecision *w = dec + c;
        w->best = best; w->type = d_type; w->order = d_order; w->prec = d_prec; w->porder = d_porder; w->method = d_method;
        w->sbps = sbp; w->shift = d_shift;
    }
    dk[c * 64 + lane] = d_k;
    if (mydbg) {
        if (lane == 0) {
            FgDebugCand *dc = &mydbg->cand[c];
            dc->wasted = wst; dc->sbps = sbp; dc->type = d_type; dc->order = d_type >= 2 ? d_order : 0;
            dc->precision = d_type == 3 ? d_prec : 0; dc->shift = d_type == 3 ? d_shift : 0;
            dc->bits = best; dc->porder = d_type >= 2 ? d_porder : 0; dc->rice_method = d_type >= 2 ? d_method : 0;
            for (uint32_t j = 0; j < FG_MAX_ORDER; j++) dc->qlp[j] = (d_type == 3 && j < d_order && j < (uint32_t)MAXO) ? bestq[c * MAXO + j] : 0;
        }
        if (d_type >= 2 && ((uint32_t)lane & ((64u >> d_porder) - 1)) == 0) mydbg->cand[c].rice_params[(uint32_t)lane >> (6 - d_porder)] = d_k;
    }
    __syncthreads();
    FG_STAMP(6);
    if (flags[0]) {
        if (tid == 0) {
            FgBlockResult *r = &results[d.out_slot];
            r->bytes = 0; r->ca = 0; r->err = FG_ERR_REDO; r->reserved = 2;
        }
        return;
    }

    // ================================================================ channel assignment (every wave, same result)
    uint32_t ca = 0, sub0 = 0, sub1 = 1;
    uint32_t bestall[4] = {0, 0, 0, 0};
#pragma unroll
    for (int cc = 0; cc < NC; cc++) bestall[cc] = rfl(dec[cc].best);
    if (MS) {
        if (d.forced_ca != 0xFF) ca = d.forced_ca;
        else {
            const uint32_t b01 = bestall[0] + bestall[1], b03 = bestall[0] + bestall[3];
            const uint32_t b13 = bestall[1] + bestall[3], b23 = bestall[2] + bestall[3];
            uint32_t mn = b01;
            if (b03 < mn) { mn = b03; ca = 1; }
            if (b13 < mn) { mn = b13; ca = 2; }
            if (b23 < mn) { mn = b23; ca = 3; }
        }
        sub0 = ca == 2 ? 3 : (ca == 3 ? 2 : 0);
        sub1 = ca == 0 ? 1 : (ca == 2 ? 1 : 3);
    }
    FG_STAMP(7);

    // ================================================================ pack: wave si < NCH writes subframe si
    const uint32_t si = c;                                    // wave index
    const bool packer = si < (uint32_t)NCH;
    const uint32_t pc = MS ? (si == 0 ? sub0 : sub1) : (packer ? si : 0);     // candidate this wave packs
    FrameBits fb;
    fb.w = (LDS uint32_t *)(ureg + (packer ? si : 0) * ((FGS_FBW + 2) * 4));
    fb.t0 = crct; fb.thi = crct + 256; fb.tlo = crct + 512; fb.t1 = crct + 768; fb.t2 = crct + 1024; fb.t3 = crct + 1280;
    fb.outw = (uint32_t *)(out + (size_t)d.out_slot * P.slot_bytes);
    fb.slot_words = P.slot_bytes / 4; fb.wbase = 0; fb.err = 0; fb.crc = 0;
    uint32_t bitpos = 0;
    if (packer) {
        for (uint32_t j = lane; j < FGS_FBW + 2; j += 64) fb.w[j] = 0;
        wave_lds_fence();
    }
    uint32_t fhdr_bits = 0;
    if (si == 0) {
    {   // frame header: assembled by lane 0 in LDS, emitted one byte per lane
        LDS uint8_t *hb = (LDS uint8_t *)hdrb;
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
        fhdr_bits = hl * 8;
        wave_lds_fence();
    }
    }
    // decision of the candidate this wave packs
    uint32_t type = 0, order = 0, sb = 0, prec = 0, po = 0, method = 0, kv = 0;
    int shift = 0;
    int32_t q[MAXO];
#pragma unroll
    for (int j = 0; j < MAXO; j++) q[j] = 0;
    if (packer) {
        type = rfl(dec[pc].type); order = rfl(dec[pc].order); sb = rfl(dec[pc].sbps); prec = rfl(dec[pc].prec);
        po = rfl(dec[pc].porder); method = rfl(dec[pc].method); shift = (int)rfl((uint32_t)dec[pc].shift);
        kv = dk[pc * 64 + lane];
        if (type == 3) {
            const int32_t qv = (lane < MAXO) ? bestq[pc * MAXO + lane] : 0;
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = (int32_t)rl((uint32_t)qv, j);
        }
        else {
            const uint32_t g = type == 2 ? order : 0;
            const int32_t c0 = g == 0 ? 0 : (int32_t)g, c1 = g < 2 ? 0 : (g == 2 ? -1 : g == 3 ? -3 : -6);
            const int32_t c2 = g < 3 ? 0 : (g == 3 ? 1 : 4), c3 = g < 4 ? 0 : -1;
#pragma unroll
            for (int j = 0; j < MAXO; j++) q[j] = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : j == 3 ? c3 : 0;
            shift = 0;
        }
    }
    const uint32_t mask = sb < 32 ? ((1u << sb) - 1) : 0xFFFFFFFFu;
    const uint32_t plen = method ? 5 : 4;
    const uint32_t lpp = 64u >> po;                                  // lanes per partition
    const uint32_t kr = type >= 2 ? (uint32_t)__shfl((int)kv, (int)((uint32_t)lane & ~(lpp - 1))) : 0;
    const bool pstart = type >= 2 && ((uint32_t)lane % lpp) == 0;
    const uint32_t skip = (type >= 2 && lane == 0) ? order : 0;      // warm-up samples are not coded
    auto pcand = [&](int32_t l, int32_t r) __attribute__((always_inline)) -> int32_t {
        if (!MS) return pc == 0 ? l : r;
        return pc == 0 ? l : pc == 1 ? r : pc == 2 ? ((l + r) >> 1) : (l - r);
    };
    // ---- the fields in front of the residual: lane 0 = subframe header byte, lanes 1..order = warm-up samples, then
    // precision/shift, coefficients, coding method + partition order
    uint32_t hpv = 0, hpb = 0, hval = 0, hvb = 0;
    {
        uint32_t hdr;
        switch (type) {
        case 0: hdr = 0x00; break;
        case 1: hdr = 0x02; break;
        case 2: hdr = 0x10 | (order << 1); break;
        default: hdr = 0x40 | ((order - 1) << 1); break;
        }
        const bool pred = type >= 2;
        const uint32_t nw = type == 0 ? 1 : (pred ? order : 0);   // sample fields
        if (lane == 0) { hpv = hdr; hpb = 8; }
        else if ((uint32_t)lane <= nw) {
            const uint32_t g = (uint32_t)lane - 1;                 // inside segment 0: order <= MAXO <= seg
            hval = (uint32_t)pcand(sL[g], (NCH == 2) ? sR[g] : 0) & mask; hvb = sb;
        }
        else if (type == 3 && (uint32_t)lane == order + 1) { hpv = prec - 1; hpb = 4; hval = (uint32_t)shift & 31; hvb = 5; }
        else if (type == 3 && (uint32_t)lane <= 2 * order + 1) { hval = (uint32_t)bestq[pc * MAXO + (lane - order - 2)] & ((1u << prec) - 1); hvb = prec; }
        else if (pred && (uint32_t)lane == (type == 3 ? 2 * order + 2 : order + 1)) { hval = (method << 4) | po; hvb = 6; }
        if (!packer) { hpb = 0; hvb = 0; }
    }
    const uint32_t hmine = hpb + hvb;
    const uint32_t hincl = wave_scan_add(hmine);
    const uint32_t htotal = rl(hincl, 63);
    // one walk over the segment; EMIT = false: returns the bit length, EMIT = true: writes the codes from bit p0 on
    auto walk_t = [&](auto VERB, auto EMIT, auto ATOM, uint32_t p0, bool inrange) __attribute__((always_inline)) -> uint32_t {
        constexpr bool verb = decltype(VERB)::value, emit = decltype(EMIT)::value, atom = decltype(ATOM)::value;
        int32_t h[MAXO];
#pragma unroll
        for (int j = 0; j < MAXO; j++) {
            int32_t x = 0;
            if (lane > 0) x = pcand(rowL[(int)seg - 1 - j - (int)rstr], (NCH == 2) ? rowR[(int)seg - 1 - j - (int)rstr] : 0);
            h[(MAXO - 1 - j) % MAXO] = x;
        }
        uint32_t pos = p0, len = 0;
        LDS uint32_t *const dummy = scrw + tid;
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
        auto step = [&](int u, uint32_t s) __attribute__((always_inline)) {
            const int32_t x = pcand(rowL[s], (NCH == 2) ? rowR[s] : 0);
            uint32_t val, vb, lead;
            if (verb) { val = (uint32_t)x & mask; vb = sb; lead = 0; }
            else {
                int32_t res;
                if (!ACC64) res = x - (fir24<MAXO>(q, h, u) >> shift);
                else res = (int32_t)((i64)x - (fir64<MAXO>(q, h, u) >> shift));
                h[u] = x;
                const uint32_t uu = ((uint32_t)res << 1) ^ (uint32_t)(res >> 31);
                lead = uu >> kr;
                val = kone | (uu & kmask);
                vb = kr + 1;
            }
            const bool coded = s >= skip;
            if (emit) put(coded ? pos + lead : pos, coded ? val : 0, coded ? vb : 0);
            const uint32_t cl_ = coded ? lead + vb : 0;
            pos += cl_; len += cl_;
        };
        uint32_t s0 = 0;
#pragma unroll 1
        for (; s0 + MAXO <= seg; s0 += MAXO) {
#pragma unroll
            for (int u = 0; u < MAXO; u++) step(u, s0 + u);
        }
#pragma unroll
        for (int u = 0; u < MAXO; u++) if (s0 + u < seg) step(u, s0 + u);
        if (emit && !atom) {
            wave_lds_fence();
            if (inrange) fb.w[cw] |= cur;
            wave_lds_fence();
        }
        return len;
    };
    auto walk = [&](bool emit, uint32_t p0, bool inrange) __attribute__((always_inline)) -> uint32_t {
        typedef std::integral_constant<bool, true> T;
        typedef std::integral_constant<bool, false> F;
        if (!emit) return type == 1 ? walk_t(T(), F(), F(), p0, inrange) : walk_t(F(), F(), F(), p0, inrange);
        if (seg < 32) return type == 1 ? walk_t(T(), T(), T(), p0, inrange) : walk_t(F(), T(), T(), p0, inrange);
        return type == 1 ? walk_t(T(), T(), F(), p0, inrange) : walk_t(F(), T(), F(), p0, inrange);
    };
    // ---- pass A: exact bit length of every lane's segment
    uint32_t mylen = 0;
    if (packer && type != 0) mylen = walk(false, 0, false);
    bool redo = __any(mylen > (1u << 24));                         // absurd code lengths: the generic kernel copes
    const uint32_t bincl = wave_scan_add(mylen);
    const uint32_t btotal = rl(bincl, 63);
    // ---- positions: wave 0 starts behind the frame header, wave 1 behind subframe 0
    if (si == 0 && lane == 0) { hand[64] = fhdr_bits + htotal + btotal; hand[68] = redo ? 1u : 0u; }
    if (si == 1 && lane == 0) hand[69] = redo ? 1u : 0u;
    __syncthreads();
    const uint32_t start = (si == 0) ? fhdr_bits : rfl(hand[64]);
    redo = rfl(hand[68]) != 0 || (NCH == 2 && rfl(hand[69]) != 0);
    const uint32_t hstart = start, bstart = start + htotal, subend = start + htotal + btotal;
    const uint32_t mystart = bstart + bincl - mylen, myend = bstart + bincl;
    bool failed = false;
    // emission of this wave's subframe; `flush_ok`: the words below the window may be written out (wave 1 must not before
    // wave 0 is through: it only fills its window, which works when the whole subframe fits)
    auto emit_subframe = [&](bool flush_ok) __attribute__((always_inline)) {
        // header fields
        if (flush_ok) fb_reserve(fb, lane, hstart, htotal);
        const uint32_t o = hstart + hincl - hmine;
        fb_or(fb, o, hpv, hpb);
        fb_or(fb, o + hpb, hval, hvb);
        wave_lds_fence();
        if (type == 0) return;
        uint32_t a = 0;
#pragma unroll 1
        while (a < 64) {
            if (flush_ok) fb_flush(fb, lane, rl(mystart, (int)a));
            const uint32_t cap = (fb.wbase << 5) + 32u * FGS_FBW - 64u;
            const uint64_t fits = __ballot((uint32_t)lane >= a && myend <= cap);
            const uint64_t shifted = fits >> a;
            const uint32_t cnt = (~shifted) ? (uint32_t)__builtin_ctzll(~shifted) : 64u - a;
            if (cnt == 0) { failed = true; break; }
            const uint32_t b = a + cnt;
            (void)walk(true, mystart, (uint32_t)lane >= a && (uint32_t)lane < b);
            wave_lds_fence();
            a = b;
        }
    };
    bool early = false;          // wave 1: subframe already in its window before wave 0 finished
    if (!redo && packer) {
        if (si == 0) {
            emit_subframe(true);
            if (NCH == 2) {
                // hand the frame over: complete words out, then the partial word and the CRC lane states
                fb_flush(fb, lane, subend);
                hand[lane] = fb.crc;
                if (lane == 0) { hand[65] = fb.w[0]; hand[66] = fb.err | (failed ? FG_ERR_REDO : 0u); }
            }
        }
        else {
            // does the whole subframe fit the window when it starts at the word of `start`?
            fb.wbase = start >> 5;
            if (subend - (fb.wbase << 5) <= 32u * FGS_FBW - 64u) { emit_subframe(false); early = true; }
        }
    }
    if (NCH == 2) __syncthreads();
    if (si != (uint32_t)(NCH - 1)) return;
    if (mydbg && lane == 0) mydbg->t[8] = clock64();                     // the last packing wave finishes the frame
    if (NCH == 2) {
        const uint32_t e0 = rfl(hand[66]);
        if (redo || (e0 & FG_ERR_REDO)) {
            if (lane == 0) { FgBlockResult *r = &results[d.out_slot]; r->bytes = 0; r->ca = 0; r->err = FG_ERR_REDO; r->reserved = 3; }
            return;
        }
        fb.err |= e0;
        fb.crc = hand[lane];
        const uint32_t tailw = rfl(hand[65]);
        if (!early) {
            // serial path: the window starts empty at wave 0's partial word
            fb.wbase = start >> 5;
            if (lane == 0) fb.w[0] = tailw;
            wave_lds_fence();
            emit_subframe(true);
        }
        else {
            if (lane == 0) fb.w[0] |= tailw;
            wave_lds_fence();
        }
    }
    else if (redo) failed = true;
    if (failed) {
        if (lane == 0) { FgBlockResult *r = &results[d.out_slot]; r->bytes = 0; r->ca = 0; r->err = FG_ERR_REDO; r->reserved = 3; }
        return;
    }
    bitpos = subend;
    // ---- zero-pad to a byte, CRC-16 over the whole frame, append
    if (bitpos & 7) bitpos += 8 - (bitpos & 7);
    fb_flush(fb, lane, bitpos);
    {
        const uint32_t nbytes = bitpos >> 3;
        const uint32_t W = nbytes >> 2, tail = nbytes & 3;
        uint32_t s = 0;
        if ((uint32_t)lane < W) s = gf16_mul(fb.crc, mult[(W - 1 - (uint32_t)lane) & 63]);
        uint32_t crc = wave_xor32(s);
        if (tail) {
            const uint32_t wv = rfl(fb.w[0]);
            for (uint32_t b = 0; b < tail; b++) crc = ((crc << 8) & 0xFFFF) ^ crct[((crc >> 8) ^ (wv >> (24 - 8 * b))) & 0xFF];
        }
        if (lane == 0) fb_or(fb, bitpos, crc, 16);
        bitpos += 16;
        wave_lds_fence();
        fb_flush(fb, lane, bitpos);
        if ((bitpos & 31) && lane == 0) {
            if (fb.wbase < fb.slot_words) fb.outw[fb.wbase] = __builtin_bswap32(fb.w[0]);
        }
        if ((bitpos & 31) && fb.wbase >= fb.slot_words) fb.err |= FG_ERR_SLOT;
    }
    if (mydbg && lane == 0) mydbg->t[9] = clock64();
    if (lane == 0) {
        FgBlockResult *r = &results[d.out_slot];
        r->bytes = bitpos >> 3; r->ca = ca; r->err = flags[1] | fb.err; r->reserved = 1;
#pragma unroll
        for (int cc = 0; cc < 4; cc++) r->best_bits[cc] = bestall[cc];
    }
#undef FG_STAMP
#undef FG_SADDR
}

}  // namespace
