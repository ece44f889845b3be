// flac_dec_fast.hip -- FLAC frame decoder for gfx950 (the common shapes), split by the two serial recurrences of the format:
//
//   fg_dec_rice_kernel     lane = frame: the bit-serial part.  Each lane walks its frame with a register bit window fed by a
//                          four-word prefetch queue, parses the subframe headers and delimits the Rice codes.  Per code it only
//                          finds the code length (count-leading-zeros + k) and parks the 32-bit window in an LDS tile; the wave
//                          then turns a whole tile of windows into residuals in parallel (quotient/remainder split, zig-zag undo)
//                          and writes it to the residual plane with coalesced stores.  The number of frames per wave is a launch
//                          parameter: a single stream has few thousand frames, so waves are kept narrow to spread the serial
//                          chains over all SIMDs; large batches use all 64 lanes.
//   fg_dec_crc_kernel      wave = frame: CRC-16 over the frame bytes, 64 lanes over interleaved 32-bit groups.
//   fg_dec_restore_kernel  lane = (frame, channel): the prediction recurrence.  Residual tiles come in through LDS with coalesced
//                          loads; fixed and LPC restoration are one FIR over a register history of up to 12 samples addressed
//                          statically (the loop is unrolled by the history length, no register moves); the finished tile is
//                          written back with the stereo decorrelation undone and the channels interleaved (or frame-planar),
//                          zeros for frames that failed (libFLAC delivers silence on a CRC mismatch).
//
// Frames using features outside these kernels (predictor order > 12, > 32-bit subframes) are flagged status 3 and redone by
// fg_decode_slow_kernel (flac_dec_kernels.hip).
//
// Reference path replaced: read_subframe_*, read_residual_partitioned_rice_, FLAC__fixed_restore_signal,
// FLAC__lpc_restore_signal, undo_channel_coding inside libFLAC.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fg_dev.h"
#include "fg_types.h"

using namespace fgdev;

#define FG_DMAXO 12
#define FG_TS 64            // residual tile of the parse kernel (samples per lane)
#define FG_TSTR 68          // its LDS row stride (words): 16-byte aligned rows, neighbouring lanes 4 banks apart
#define FG_DEC_RPARAMS 256  // Rice parameters kept per subframe for FLAC__Frame.subframes[] (partition order <= 8)

#define FG_LDSP __attribute__((address_space(3)))

extern "C" int fg_func_set_lds(const void *fn, size_t bytes);   // fg_ctx.cpp: per device, thread-safe

namespace {

__device__ __forceinline__ uint32_t be32(uint32_t v) { return __builtin_bswap32(v); }

__device__ __forceinline__ uint32_t wave_max32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o); v = t > v ? t : v; }
    return v;
}

// MSB-first bit reader.  The window is w0, w1 with w2 and w3 (just requested from LDS) queued behind; `s` in
// [0, 31] is 32 minus the bit offset into w0 (offset 1..32), so peek() is one v_alignbit and always returns 32 valid
// bits, and consuming up to 32 bits advances by at most one word.  The stream reaches the lane through a private ring
// of 16-byte groups in LDS:
//   * HBM -> registers: every lane loads the next few aligned groups of its own frame at the start of a residual tile
//     (issue()), with wave-uniform control flow, and parks them in the ring one tile later (land()) -- the memory
//     latency is covered by a whole tile of parsing and never sits inside the per-code loop.  (A row-by-row refill --
//     64 lanes load 64 consecutive groups of ONE frame, coalesced -- was tried and is slower: the scalar bookkeeping
//     per visited row costs more than the eight divergent loads it saves.)
//   * ring -> window: one ds_read per 32 bits consumed, one word ahead of its use.  The ring holds the words already
//     byte-swapped (big-endian stream -> register order, done once per group when it is parked), and its first FG_RMIR
//     groups are repeated behind its end, so that the per-code loop of a tile (at most 256 bytes further) reads on
//     without wrapping its address.
// Groups are aligned to 16 bytes in memory, so a load never straddles a page and the group that holds the last stream
// byte is the last one touched (indices are clamped to it).  consume() checks that the ring holds the next word and
// fetches synchronously if not (headers, escapes, very long codes); consume_fast() relies on the tile-start
// guarantee of FG_RAHEAD groups, enough for a tile of codes of at most 32 bits.
typedef uint32_t fg_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) fg_u32x4 *FgGroupPtr;
#define FG_RG 64                      // ring capacity in groups (1 KiB per lane)
#define FG_RMIR 20                    // groups repeated behind the end (a tile's 256 bytes + the look-ahead words)
#define FG_RSTR ((FG_RG + FG_RMIR) * 4)   // ring row stride in words
#define FG_RAHEAD 20                  // groups guaranteed ahead of the read position at a tile start
#define FG_RCAP 52                    // never hold more than this many groups ahead
#define FG_PF 8                       // groups fetched per lane and tile

struct BitRdState { uint32_t w0, w1, w2, w3, s, wb; };

struct BitRd {
    FgGroupPtr fg;          // group that holds the first byte of the frame
    uint32_t glim;          // last loadable group index relative to fg
    uint32_t skip0;         // bit offset of the frame start inside fg[0]
    uint32_t *ring;         // this lane's ring (LDS, 512-byte aligned)
    uint32_t w0, w1, w2, w3, s;
    uint32_t wb;            // byte offset (relative to fg) of the next word to read from the ring; w3 = word wb/4 - 1
    uint32_t H;             // groups below H are in the ring (or already consumed)
    uint32_t hlim;          // this reader must not fetch groups at or beyond hlim itself (fused kernel: another wave may still
                            // be parking the groups 64 below them into the same slots, see FgRingFeed); ~0 = no limit
    bool over;              // ... it wanted to: the frame goes to the generic decoder
    uint32_t pfH, pfn;      // groups [pfH, pfH + FG_PF) are in flight, the first pfn of them count
    bool pfvalid;
    fg_u32x4 pf[FG_PF];

    __device__ __forceinline__ fg_u32x4 ldgroup(uint32_t g) const { return fg[g < glim ? g : glim]; }
    // group g of the frame into its ring slot (and the slot's repeat behind the end), words in register order
    __device__ __forceinline__ void park(uint32_t g, fg_u32x4 v)
    {
        v.x = be32(v.x); v.y = be32(v.y); v.z = be32(v.z); v.w = be32(v.w);
        const uint32_t slot = g & (FG_RG - 1);
        *(fg_u32x4 *)&ring[slot * 4] = v;
        if (slot < FG_RMIR) *(fg_u32x4 *)&ring[(slot + FG_RG) * 4] = v;
    }
    __device__ __forceinline__ void selfload()
    {
        if (H >= hlim) { over = true; H++; return; }       // (what is read from here on is unspecified; the caller gives the frame up)
        park(H, ldgroup(H));
        H++;
    }
    __device__ __forceinline__ uint32_t ringword() const { return *(const uint32_t *)((const char *)ring + (wb & (FG_RG * 16 - 1))); }
    __device__ __forceinline__ void save(BitRdState &t) const { t.w0 = w0; t.w1 = w1; t.w2 = w2; t.w3 = w3; t.s = s; t.wb = wb; }
    __device__ __forceinline__ void restore(const BitRdState &t) { w0 = t.w0; w1 = t.w1; w2 = t.w2; w3 = t.w3; s = t.s; wb = t.wb; }
    __device__ __forceinline__ uint32_t fetch()
    {
        while ((wb >> 4) >= H) selfload();
        const uint32_t x = ringword();
        wb += 4;
        return x;
    }
    // where the reading starts (init_pos), then -- the ring may have been filled meanwhile -- the window (init_words)
    __device__ __forceinline__ void init_pos(FgGroupPtr frame_group, uint32_t group_limit, uint32_t frame_bit0, uint32_t start_bit, uint32_t *lds_ring)
    {
        fg = frame_group; glim = group_limit; skip0 = frame_bit0; ring = lds_ring;
        const uint32_t b = frame_bit0 + start_bit;
        const uint32_t w = b >> 5;
        wb = w * 4; H = w >> 2; pfH = H; pfn = 0; pfvalid = false;
        s = b & 31;             // (bit offset into the first word until init_words)
    }
    __device__ __forceinline__ void init_words()
    {
        const uint32_t sk = s;
        w0 = 0;
        if (sk) w0 = fetch();
        w1 = fetch();
        w2 = fetch();
        w3 = fetch();
        s = (32 - sk) & 31;
    }
    // bits consumed since the frame start
    __device__ __forceinline__ uint32_t pos() const { return 8u * wb - 96u - s - skip0; }
    __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(w0, w1, s); }
    __device__ __forceinline__ void consume(uint32_t n)   // n <= 32
    {
        s -= n;
        if ((int32_t)s < 0) { s += 32; w0 = w1; w1 = w2; w2 = w3; w3 = fetch(); }
    }
    // straight-line variant for the per-code loop (no availability check, see FG_RAHEAD)
    __device__ __forceinline__ void consume_fast(uint32_t n)
    {
        s -= n;
        const bool adv = (int32_t)s < 0;
        s &= 31;
        w0 = adv ? w1 : w0;
        w1 = adv ? w2 : w1;
        if (adv) { w2 = w3; w3 = ringword(); wb += 4; }
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
            if (pos() > limit_bits) return z;
        }
    }
    // tile start, step 1: park the groups requested one tile ago.  Every lane parks all FG_PF registers: ring slots of
    // groups that are not valid yet are free, and a group that is already there is rewritten with the same bytes.
    __device__ __forceinline__ void land()
    {
        if (fg && pfvalid) {       // lanes without a frame own no ring; nothing was requested before the first tile
            // only groups that are not there yet, and only into slots whose old group (64 below) lies behind the read
            // position: after a leap -- a code of several hundred bytes, fetched word by word -- the registers hold
            // groups the reader has long passed, and their slots belong to newer ones
            const uint32_t cg = wb >> 4;
            if (pfH >= H && pfH + FG_PF <= cg + 63) {       // (the usual case, straight-line)
#pragma unroll
                for (int t = 0; t < FG_PF; t++) park(pfH + t, pf[t]);
            }
            else {
#pragma unroll
                for (int t = 0; t < FG_PF; t++) { const uint32_t g = pfH + t; if (g >= H && g < cg + 63) park(g, pf[t]); }
            }
        }
        const uint32_t h2 = pfH + pfn;
        H = h2 > H ? h2 : H;
        pfn = 0;
    }
    // tile start, step 2: guarantee the look-ahead of the lanes that will parse, then request the next FG_PF groups
    // (straight-line code: one address, FG_PF loads; a lane that is far enough ahead simply does not count them.  A
    // request count that follows the consumption -- fewer loads and parks on most tiles -- was tried: the conditional
    // loads cost more in waits than they save.)
    __device__ __forceinline__ void ensure_ahead(bool on)
    {
        const uint32_t cg = wb >> 4;
        if (__any(on && H < cg + FG_RAHEAD)) { if (on) while (H < cg + FG_RAHEAD) selfload(); }
    }
    __device__ __forceinline__ void issue(bool on, bool ensure = true)
    {
        const uint32_t cg = wb >> 4;
        if (ensure) ensure_ahead(on);
        // start of the batch, pulled back at the very end of the stream so that every load stays inside it
        const uint32_t last = glim >= FG_PF - 1 ? glim - (FG_PF - 1) : 0;
        pfH = H < last ? H : last;
        pfn = (on && H + FG_PF <= cg + FG_RCAP) ? FG_PF : 0;
        if (!fg) return;
        pfvalid = true;
        if (glim >= FG_PF - 1) {
            const FgGroupPtr src = fg + pfH;
#pragma unroll
            for (int t = 0; t < FG_PF; t++) pf[t] = src[t];
        }
        else {
#pragma unroll
            for (int t = 0; t < FG_PF; t++) pf[t] = ldgroup(pfH + t);
        }
    }
};

__device__ __forceinline__ int32_t unzig(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1); }

typedef uint32_t fg_crc_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
__global__ void __launch_bounds__(256)
fg_dec_crc_kernel(const uint8_t *stream, const FgDecFrame *frames, uint32_t nframes, FgDecResult *results, const uint16_t *crctab,
                  const u64 *offsets, u64 stream_len)
{
    (void)crctab;
    const int lane = threadIdx.x & 63;
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    // (offsets: the pass starts from the frame positions alone, beside the header pass -- FgDecSelf; a frame the header pass
    // turns down is checked for nothing)
    uint32_t fb;
    u64 boff;
    if (offsets) {
        // (agent-scope loads, as in the header pass)
        const u64 o0 = __hip_atomic_load(&offsets[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), o1 = __hip_atomic_load(&offsets[f + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool inside = o0 < stream_len && o1 <= stream_len && o1 > o0 && o1 - o0 < 0x7FFFFFFFull;
        fb = inside ? (uint32_t)(o1 - o0) : 0; boff = inside ? o0 : 0;
    }
    else { fb = frames[f].bytes; boff = frames[f].byte_off; }
    if (fb < 3) { if (offsets && lane == 0) results[f].crc = 0; return; }      // (on its own the pass owns the word: the header pass leaves it alone)
    const uint8_t *fp = stream + boff;
    const uint32_t nbytes = fb - 2;
    // from the first 4-byte aligned address: head bytes serially, then interleaved granules of four words, then what is left
    const uint32_t mis = (uint32_t)((uintptr_t)fp & 3);
    const uint32_t head = mis ? (4 - mis) : 0;
    const uint32_t hb = head < nbytes ? head : nbytes;
    const uint32_t W = (nbytes - hb) >> 2, tail = (nbytes - hb) & 3;
    const uint32_t G = W >> 2, Wr = W & 3;
    const uint32_t *wptr = (const uint32_t *)(fp + hb);
    // crc(head || body) = crc(head) * x^(128 G) + crc(body): the head's remainder takes the place of a granule in front of granule 0
    // (lane pad - 1 of the first step; a step more when the granules fill their first step) and rides through the same folds
    uint32_t hcrc = 0;
    for (uint32_t b = 0; b < hb; b++) hcrc = crc16_byte(hcrc, fp[b]);
    const uint32_t pad0 = (64 - (G & 63)) & 63;
    const uint32_t pad = (hb && G && pad0 == 0) ? 64u : pad0, T = (G + pad) >> 6;
    uint32_t s = 0;
    for (uint32_t t = 0; t < T; t++) {
        const int qi = (int)(t * 64 + lane) - (int)pad;
        fg_crc_u32x4 g = {0, 0, 0, 0};
        if (qi >= 0) g = *(const fg_crc_u32x4 *)(wptr + 4 * qi);
        s = crc16_mul_x8192(s);                                   // x^8192: the 64 granules of a step
        uint32_t c = crc16_word(0, be32(g.x));
        c = crc16_word(c, be32(g.y)); c = crc16_word(c, be32(g.z)); c = crc16_word(c, be32(g.w));
        if (qi == -1) c = hcrc;
        s ^= c;
    }
    // x^(128 (63 - lane)): the granules behind this lane's last one
    static const uint16_t fold[64] = {
        0xB7B3, 0x9259, 0x831B, 0x0105, 0xDB58, 0x6481, 0xF3CE, 0x3964, 0xF11D, 0xCACD, 0x5C80, 0xB2F1, 0xA1F5, 0xB59F, 0x348E, 0x033E,
        0x1164, 0x127C, 0x937A, 0xC821, 0x2CA3, 0x5F6E, 0xE609, 0x9717, 0xDDD9, 0x015A, 0x33A8, 0x125E, 0x27C2, 0x934F, 0x1072, 0x8115,
        0x136A, 0x0013, 0x4831, 0xE491, 0x3BFC, 0x5DF6, 0x4AE2, 0x1738, 0x9661, 0x25CA, 0xB797, 0x1056, 0x031A, 0x936B, 0x927D, 0x0114,
        0x8104, 0x4936, 0x2DA4, 0x965B, 0x4BAE, 0x814F, 0x1674, 0x0016, 0xA5DF, 0x924B, 0x021E, 0x8107, 0x926F, 0x8011, 0x0106, 0x0001};
    if (T) s = gf16_mul(s, fold[lane]);
    const uint32_t body = wave_xor32(s);
    uint32_t crc = G ? body : hcrc;
    for (uint32_t k = 0; k < Wr; k++) crc = crc16_word(crc, be32(wptr[4 * G + k]));
    for (uint32_t b = 0; b < tail; b++) crc = crc16_byte(crc, fp[hb + W * 4 + b]);
    const uint32_t stored = ((uint32_t)fp[nbytes] << 8) | fp[nbytes + 1];
    if (lane == 0) {
        // runs beside the parse kernel (which owns `err`): the mismatch travels in bit 31, the restore kernel merges it
        results[f].crc = crc | (crc != stored ? 0x80000000u : 0u);
    }
}


// The first samples of every subframe (the predictor's warm-up, or the start of a verbatim subframe) as they were coded:
// FLAC__Subframe_Fixed / _LPC.warmup of the frame handed to the write callback.  thread = (subframe, j).
__global__ void __launch_bounds__(256)
fg_dec_warmup_kernel(const FgDecFrame *frames, uint32_t nframes, uint32_t C, const FgDecSub *subs, const int32_t *scratch, int32_t *warm)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t sf = t >> 5, j = t & 31;
    if (sf >= nframes * C) return;
    const uint32_t f = sf / C, ch = sf % C;
    const FgDecFrame fr = frames[f];
    int32_t v = 0;
    if (fr.bytes != 0 && fr.channels == C && (subs[sf].flags & (1u << 12)) && j < subs[sf].order && j < fr.n)
        v = scratch[fr.out_off * C + (u64)ch * fr.n + j];
    warm[(size_t)sf * 32 + j] = v;
}

}  // namespace

extern "C" int fg_launch_decode_warmup(const FgDecFrame *d_frames, uint32_t nframes, uint32_t channels, const FgDecSub *d_subs,
                                       const int32_t *d_scratch, int32_t *d_warm, hipStream_t stream)
{
    if (nframes == 0) return 0;
    const uint32_t total = nframes * channels * 32;
    hipLaunchKernelGGL(fg_dec_warmup_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, d_frames, nframes, channels, d_subs, d_scratch, d_warm);
    return (int)hipGetLastError();
}

// (The lane-serial decoders of rounds 1 and 2 -- fg_dec_rice_kernel + fg_dec_restore_kernel, fg_dec_fused_kernel -- left the tree in
// round 5, their entry points and fg_dec_fix_kernel in round 6; the wave-parallel parser and its restore kernel, flac_dec_wave.hip, replace them.)
// CRC-16 of every frame; independent of the parse kernel, so the caller may run it on a second stream beside it
extern "C" int fg_launch_decode_crc(const uint8_t *d_stream, const FgDecFrame *d_frames, uint32_t nframes, FgDecResult *d_results,
                                    const uint16_t *d_crctab, hipStream_t stream, const unsigned long long *d_offsets, unsigned long long stream_len)
{
    if (nframes == 0) return 0;
    hipLaunchKernelGGL(fg_dec_crc_kernel, dim3((nframes + 3) / 4), dim3(256), 0, stream, d_stream, d_frames, nframes, d_results, d_crctab,
                       (const u64 *)d_offsets, (u64)stream_len);
    return (int)hipGetLastError();
}
