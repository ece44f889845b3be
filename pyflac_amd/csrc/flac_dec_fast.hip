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
#ifdef FG_LEGACY        // (make LEGACY=1) round 1 / round 2 parse kernels: fg_dec_rice_kernel, and the walk the fused kernel shares with it
// ------------------------------------------------------------------------------------------------ parse
// Tile protocol (LDS): tile[row][col] holds either a finished value or the 32-bit window at the start of a Rice code
// ("window form": leading zeros = quotient, then the stop bit, then k remainder bits).  rowk[row] = k when the tile is in
// window form, 0xFF when every entry is a finished value, 0xFE when the subframe is CONSTANT (value in rowc[row]);
// a 64-bit mask marks entries of a window-form tile that are finished values anyway.  The per-row facts stay in the
// registers of the owning lane; the flush reads them with v_readlane (rows are visited in wave-uniform order).
// Two waves per group of G frames.  Wave 0 parses (lane = frame, strictly serial per lane); wave 1 converts the tile wave 0
// finished one step earlier into residuals and stores them (the parallel part), so that it is off the serial path.  Tiles
// are double-buffered in LDS, the per-row facts of a tile travel through `meta`, one workgroup barrier per tile.
#define FG_META 8                     // words per row and tile: (rn << 8 | tk), offset lo/hi, mask lo/hi, constant value

__device__ __forceinline__ void fg_dec_flush_tile(const uint32_t *tile, const uint32_t *meta, uint32_t G, uint32_t i0, int lane, int32_t *scratch)
{
    const uint32_t icol = i0 + (uint32_t)lane;
    uint32_t special = 0;
    if ((uint32_t)lane < G) {
        const uint32_t m = meta[lane * FG_META], rn_l = m >> 8, tk_l = m & 0xFF;
        special = rn_l > i0 && (tk_l >= 0xFE || (meta[lane * FG_META + 3] | meta[lane * FG_META + 4]) != 0);
    }
    if (!__any(special)) {
        // common case: every row is a full tile of code windows.  Four rows per pass: 16 lanes per row, four entries per
        // lane (one 16-byte LDS read, one 16-byte store)
        const uint32_t rsub = (uint32_t)lane >> 4, q4 = ((uint32_t)lane & 15) * 4;
        for (uint32_t r0 = 0; r0 < G; r0 += 4) {
            const uint32_t r = r0 + rsub;
            const uint32_t src = r < G ? r : 0;
            const uint32_t m = meta[src * FG_META];
            const uint32_t rn_r = r < G ? (m >> 8) : 0, kk = m & 0xFF;
            const u64 off_r = ((u64)meta[src * FG_META + 2] << 32) | meta[src * FG_META + 1];
            if (i0 + q4 < rn_r) {
                const uint4 pw = *(const uint4 *)&tile[r * FG_TSTR + q4];
                const uint32_t p4[4] = {pw.x, pw.y, pw.z, pw.w};
                int32_t res[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t lz = (uint32_t)__clz(p4[e]);
                    const uint32_t rest = (p4[e] << lz) << 1;
                    const uint32_t u = (lz << kk) | (kk ? (rest >> (32 - kk)) : 0);
                    res[e] = unzig(u);
                }
                int32_t *dst = scratch + off_r + i0 + q4;
                if ((((uintptr_t)dst) & 15) == 0) *(int4 *)dst = make_int4(res[0], res[1], res[2], res[3]);
                else { dst[0] = res[0]; dst[1] = res[1]; dst[2] = res[2]; dst[3] = res[3]; }
            }
        }
        return;
    }
    for (uint32_t r = 0; r < G; r++) {
        const uint32_t m = meta[r * FG_META], rn_s = m >> 8, kk = m & 0xFF;
        if (i0 >= rn_s) continue;
        const u64 off_s = ((u64)meta[r * FG_META + 2] << 32) | meta[r * FG_META + 1];
        uint32_t val;
        if (kk == 0xFE) val = meta[r * FG_META + 5];
        else {
            val = tile[r * FG_TSTR + lane];
            if (kk != 0xFF) {
                const uint32_t mlo = meta[r * FG_META + 3], mhi = meta[r * FG_META + 4];
                const uint32_t done = ((lane < 32 ? mlo : mhi) >> (lane & 31)) & 1;
                const uint32_t lz = (uint32_t)__clz(val);
                const uint32_t rest = (val << lz) << 1;
                const uint32_t u = (lz << kk) | (kk ? (rest >> (32 - kk)) : 0);
                val = done ? val : (uint32_t)unzig(u);
            }
        }
        if (icol < rn_s) scratch[off_s + icol] = (int32_t)val;
    }
}

// LDS areas of a parse workgroup.  subp / frm exist only in the fused kernel: subp[ch & 3][row][16] = order, shift, wasted,
// valid, q[12] of the subframe being parsed; frm[row][8] = n, channels, channel assignment, out_off lo / hi, accepted.
struct FgParseLds { uint32_t *rings, *tiles, *metas, *ctrl, *subp, *frm, *feed; };
#define FG_SUBP 16
#define FG_FRM 8

// The reader of frame `fr` positioned behind its header (window not loaded yet): the same arithmetic for the parser and for
// the wave that feeds its ring.
__device__ __forceinline__ void fg_frame_reader(BitRd &br, const uint8_t *stream, u64 stream_len, const FgDecFrame &fr, uint32_t *ring)
{
    const uintptr_t sa = (uintptr_t)stream;
    const FgGroupPtr gbase = (FgGroupPtr)(sa & ~(uintptr_t)15);
    const u64 mis = (u64)(sa & 15);
    const u64 total_groups = (mis + stream_len + 15) >> 4;
    const u64 fb = mis + fr.byte_off;
    const u64 g0 = fb >> 4;
    const u64 room = total_groups > g0 ? total_groups - g0 - 1 : 0;
    br.init_pos(gbase + g0, room > 0x0FFFFFF0ull ? 0x0FFFFFF0u : (uint32_t)room, (uint32_t)(fb & 15) * 8, fr.hdr_bytes * 8, ring);
}

// The parser wave (lane = frame): shared by fg_dec_rice_kernel and fg_dec_fused_kernel.
// In the fused kernel the ring is fed by the converter wave (FgRingFeed below): the parser only publishes its read position
// (and how far it fetched for itself) at the end of a tile and picks up how far the ring is filled at the start of the next.
template <bool FUSED>
__device__ __forceinline__ void fg_parse_wave(const uint8_t *stream, u64 stream_len, const FgDecFrame *frames, uint32_t nframes, uint32_t G,
                                              uint32_t narrow, FgDecSub *subs, FgDecResult *results, u64 *prof, uint16_t *rparams,
                                              const FgParseLds L, const int lane)
{
    uint32_t *const rings = L.rings, *const tiles = L.tiles, *const metas = L.metas, *const ctrl = L.ctrl;
    uint32_t it = 0;                                     // tiles finished so far: buffer it & 1 is the one being filled
#define tile (tiles + (it & 1) * G * FG_TSTR)
    const uint32_t f = blockIdx.x * G + lane;
    const bool mine = (uint32_t)lane < G && f < nframes;
    FgDecFrame fr;
    fr.byte_off = 0; fr.out_off = 0; fr.bytes = 0; fr.n = 0; fr.hdr_bytes = 0; fr.channels = 0; fr.ca = 0; fr.bps = 0;
    if (mine) fr = frames[f];
    const bool accepted = mine && fr.bytes != 0;          // the header pass rejects frames by zeroing `bytes`
    bool alive = accepted;
    uint32_t err = 0;
    if (alive && fr.bytes < fr.hdr_bytes + 2) { err = 1; alive = false; }
    const uint32_t n = fr.n, C = fr.channels;
    const uint32_t Cmax = wave_max32(alive ? C : 0), nmax = wave_max32(alive ? n : 0);
    if (lane == 0) { ctrl[1] = (nmax + FG_TS - 1) / FG_TS; ctrl[0] = Cmax * ((nmax + FG_TS - 1) / FG_TS); }
    if (FUSED && (uint32_t)lane < G) {
        uint32_t *fm = L.frm + lane * FG_FRM;
        fm[0] = alive ? n : 0; fm[1] = C; fm[2] = fr.ca; fm[3] = (uint32_t)fr.out_off; fm[4] = (uint32_t)(fr.out_off >> 32); fm[5] = alive ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t end_bits = alive ? (fr.bytes - 2) * 8 : 0;
    BitRd br;
    br.fg = nullptr; br.glim = 0; br.skip0 = 0; br.ring = rings; br.w0 = 0; br.w1 = 0; br.w2 = 0; br.w3 = 0; br.s = 0; br.wb = 12; br.H = 0;
    br.hlim = ~0u; br.over = false;
    br.pfH = 0; br.pfn = 0; br.pfvalid = false;
    if (alive) fg_frame_reader(br, stream, stream_len, fr, rings + lane * FG_RSTR);
    // Read positions (in groups) at the end of the last tile and of the one before.  What the feeding wave parks during a
    // tile was requested one tile earlier behind the position of two tiles ago (cg_e2), in [cg_e2, cg_e2 + 60): a group this
    // wave fetched for itself at or beyond cg_e2 + 64 would share a slot with one of those and could be overwritten by
    // it.  Tiles of ordinary codes advance 16 groups at most; a frame with codes of hundreds of bytes (residuals near
    // 2^31 under a small Rice parameter) leaps further, hits the limit and goes to the generic decoder (status 3).
    uint32_t cg_e1 = br.wb >> 4, cg_e2 = cg_e1;
    if (FUSED) br.hlim = cg_e2 + 64;
    if (FUSED) {
        __syncthreads();                                    // the feeding wave has filled the first groups of every ring
        if (alive) { const uint32_t hp = L.feed[128 + lane]; br.H = hp > br.H ? hp : br.H; }
    }
    if (alive) br.init_words();

    u64 tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = prof ? clock64() : 0;
#define FG_TICK(i) do { if (prof) { const u64 now_ = clock64(); tp[i] += now_ - tlast; tlast = now_; } } while (0)
    FG_TICK(0);
    for (uint32_t ch = 0; ch < Cmax; ch++) {
        // ---------------- subframe header (per lane)
        uint32_t kind = 0;              // 0 idle (constant / dead), 1 Rice-coded residual, 2 verbatim
        uint32_t order = 0, sb = 0, k = 0, raw = 0, po = 0, psz = 0, plen = 4, escv = 15, pend = 0, part = 0;
        u64 mask0 = 0;
        uint32_t cval = 0;
        bool is_esc = false, aligned = false, isconst = false;
        const bool on = alive && ch < C;
        if (FUSED && !on && (uint32_t)lane < G) {
            uint32_t *sp = L.subp + ((ch & 3) * G + lane) * FG_SUBP;
            for (uint32_t j = 0; j < FG_SUBP; j++) sp[j] = 0;
        }
        if (on) {
            FgDecSub *sd = &subs[(size_t)f * C + ch];
            sb = fr.bps;
            if ((fr.ca == 1 && ch == 1) || (fr.ca == 2 && ch == 0) || (fr.ca == 3 && ch == 1)) sb++;
            const uint32_t hdr = br.bits(8);
            uint32_t wasted = 0;
            if (hdr & 0x80) err = 1;
            if (!err && (hdr & 1)) { wasted = br.unary(end_bits) + 1; if (wasted >= sb) err = 1; else sb -= wasted; }
            if (!err && sb > (narrow ? 24u : 32u)) err = 3;
            const uint32_t t = (hdr >> 1) & 0x3F;
            uint32_t mode = 0;
            if (t == 0) mode = 0;
            else if (t == 1) mode = 1;
            else if (t >= 8 && t <= 12) { mode = 2; order = t & 7; }
            else if (t >= 32) { mode = 2; order = (t & 31) + 1; }
            else if (!err) err = 1;
            if (!err && order > n) err = 1;
            if (!err && order > FG_DMAXO) err = 3;
            int shift = 0;
            uint32_t sprec = 0;
            if (!err) {
                if (mode == 0) { isconst = true; cval = (uint32_t)br.sbits(sb); order = 0; }
                else if (mode == 1) { kind = 2; order = 0; }
                else {
                    for (uint32_t i = 0; i < order; i++) tile[lane * FG_TSTR + i] = (uint32_t)br.sbits(sb);
                    mask0 = ((u64)1 << order) - 1;      // order <= 12 here
                    if (t >= 32) {
                        const uint32_t prec = br.bits(4) + 1;
                        sprec = prec;
                        if (prec == 16) err = 1;
                        shift = br.sbits(5);
                        if (shift < 0) err = 1;
                        // the 32-bit restore is exact only under libFLAC's own width rule (lpc.c: bps + precision + ilog2(order) <= 32)
                        if (!err && narrow && sb + prec + ilog2_32(order) > 32) err = 3;
                        if (!err) for (uint32_t j = 0; j < order; j++) sd->q[j] = br.sbits(prec);
                    }
                    else {
                        // fixed predictor of order k as FIR with binomial coefficients
                        const int32_t c0 = (int32_t)order, c1 = order < 2 ? 0 : (order == 2 ? -1 : order == 3 ? -3 : -6);
                        const int32_t c2 = order < 3 ? 0 : (order == 3 ? 1 : 4), c3 = order < 4 ? 0 : -1;
                        sd->q[0] = c0; sd->q[1] = c1; sd->q[2] = c2; sd->q[3] = c3;
                    }
                    if (!err) {
                        const uint32_t method = br.bits(2);
                        if (method > 1) err = 1;
                        po = br.bits(4);
                        plen = method ? 5 : 4; escv = method ? 31 : 15;
                        psz = n >> po;
                        if ((po > 0 && ((n & ((1u << po) - 1)) != 0 || psz < order)) || (po == 0 && n < order)) err = 1;
                        aligned = po == 0 || (psz % FG_TS) == 0;
                        pend = order;
                        kind = 1;
                    }
                }
            }
            if (err) { alive = false; kind = 0; isconst = false; }
            else {
                // what FLAC__Frame.subframes[] reports (fg_types.h FgDecSub): type, coefficient precision, partition order, method
                const uint32_t stype = mode == 0 ? 0u : mode == 1 ? 1u : (t >= 32 ? 3u : 2u);
                sd->order = order; sd->shift = shift; sd->wasted = wasted;
                sd->flags = stype | (sprec << 2) | (po << 7) | ((plen == 5 ? 1u : 0u) << 11) | (1u << 12);
                if (mode == 0) sd->q[0] = (int32_t)cval;
            }
            if (FUSED) {
                // what the recurrence and the output waves need, two / three tiles from now (four buffers: a channel may be a
                // single tile long)
                uint32_t *sp = L.subp + ((ch & 3) * G + lane) * FG_SUBP;
                sp[0] = err ? 0u : order; sp[1] = err ? 0u : (uint32_t)shift; sp[2] = err ? 0u : wasted; sp[3] = err ? 0u : 1u;
                for (uint32_t j = 0; j < 12; j++) sp[4 + j] = (!err && j < order && mode == 2) ? (uint32_t)sd->q[j] : 0u;
            }
        }
        uint32_t rn = (on && !err) ? n : 0;                       // samples this lane's row contributes
        const u64 roff = fr.out_off * C + (u64)ch * n;
        const uint32_t start = order;
        FG_TICK(1);

        // ---------------- residual tiles
        for (uint32_t i0 = 0; i0 < nmax; i0 += FG_TS) {
            const bool act = alive && kind != 0 && i0 < n;
            uint32_t tk = 0xFF;
            u64 tmask = (i0 == 0) ? mask0 : 0;
            bool fastlane = false;
            if (act && kind == 1 && aligned && i0 > 0 && i0 + FG_TS <= n) {
                while (i0 >= pend) {
                    k = br.bits(plen);
                    is_esc = (k == escv);
                    if (is_esc) raw = br.bits(5);
                    if (rparams && part < FG_DEC_RPARAMS) rparams[((size_t)f * C + ch) * FG_DEC_RPARAMS + part] = (uint16_t)(is_esc ? (0x8000u | (raw << 8)) : k);
                    part++;
                    pend = po == 0 ? n : part * psz;
                }
                fastlane = !is_esc;
            }
            if (__any(act)) {
                if (FUSED) {
                    // how far the feeding wave has got (published a tile ago); a lane that is short all the same -- very
                    // long codes, the first tiles of a frame -- fetches for itself
                    if (alive) { const uint32_t hp = L.feed[128 + lane]; br.H = hp > br.H ? hp : br.H; }
                    br.hlim = cg_e2 + 64;
                    br.ensure_ahead(act);
                }
                else {
                    br.land();
                    br.issue(act);
                }
                FG_TICK(2);
                if (act) {
                    uint32_t *row = &tile[lane * FG_TSTR];
                    uint32_t ii = 0;
                    if (!__any(!fastlane)) {
                        // every parsing lane sits inside one Rice partition for the whole tile: delimit the codes only
                        // (straight-line code).  A code that does not fit the window (rare) voids the attempt: the
                        // reader is rewound and the general loop below decodes the tile.
                        BitRdState keep;
                        br.save(keep);
                        // Per code: lz = leading zeros of the window, length = lz + k + 1.  The dependent chain is kept
                        // short: the word advance is decided by comparing lz with (s - k - 1), prepared one code earlier.
                        // The window here is w0, w1 with one word looked ahead: `addr` is the LDS address of that word's
                        // ring slot, it moves on by a word with every advance and the slot is simply read again after every
                        // code (the same word when nothing moved) -- selects instead of an EXEC-mask region, which costs
                        // more.  The 64 codes are one block of assembly so that the order is ours: the ring read is issued
                        // as soon as the advance is known and its result is used by the last instruction of the NEXT code
                        // (two registers take the reads in turn), a code and a half of work over the LDS latency; the four
                        // windows of a group sit in v[248:251] and leave with one ds_write_b128.
                        const uint32_t kp1 = k + 1;
                        uint32_t w0 = br.w0, w1 = br.w1, na = br.w2, nb, tsh = br.s, pmin = 0xFFFFFFFFu, t_sm, t_lz, t_a4;
                        int32_t smk = (int32_t)(br.s - kp1);
                        const uint32_t a0 = (uint32_t)(uintptr_t)(const FG_LDSP char *)((const char *)br.ring + ((br.wb - 8) & (FG_RG * 16 - 1)));
                        uint32_t addr = a0;
                        const uint32_t rowa = (uint32_t)(uintptr_t)(FG_LDSP uint32_t *)row;
#define FG_RC(P, A, B, N, EXTRA)                                                  \
    "v_alignbit_b32 " P ", %[w0], %[w1], %[t]\n"                                  \
    "v_ffbh_u32 %[lz], " P "\n"                                                   \
    "v_cmp_lt_i32 vcc, %[smk], %[lz]\n"                                           \
    "v_cndmask_b32_e64 %[a4], 0, 4, vcc\n"                                        \
    "v_add_u32 %[addr], %[addr], %[a4]\n"                                         \
    "ds_read_b32 " B ", %[addr]\n" EXTRA                                          \
    "v_sub_u32 %[t], %[smk], %[lz]\n"                                             \
    "v_and_b32 %[sm], 31, %[t]\n"                                                 \
    "v_sub_u32 %[smk], %[sm], %[kp1]\n"                                           \
    "v_cndmask_b32 %[w0], %[w0], %[w1], vcc\n"                                    \
    "s_waitcnt lgkmcnt(" #N ")\n"                                                 \
    "v_cndmask_b32 %[w1], %[w1], " A ", vcc\n"
#define FG_WST(OFF) "ds_write_b128 %[row], v[248:251] offset:" #OFF "\n"
#define FG_RG4(OFF)                                                                                          \
    FG_RC("v248", "%[na]", "%[nb]", 2, "")                                                                   \
    FG_RC("v249", "%[nb]", "%[na]", 1, "v_min3_u32 %[pmin], %[pmin], v248, v249\n")                          \
    FG_RC("v250", "%[na]", "%[nb]", 1, "")                                                                   \
    FG_RC("v251", "%[nb]", "%[na]", 2, FG_WST(OFF) "v_min3_u32 %[pmin], %[pmin], v250, v251\n")
                        asm volatile(FG_RG4(0) FG_RG4(16) FG_RG4(32) FG_RG4(48) FG_RG4(64) FG_RG4(80) FG_RG4(96) FG_RG4(112)
                                     FG_RG4(128) FG_RG4(144) FG_RG4(160) FG_RG4(176) FG_RG4(192) FG_RG4(208) FG_RG4(224) FG_RG4(240)
                                     "s_waitcnt lgkmcnt(0)\n"
                                     : [w0] "+v"(w0), [w1] "+v"(w1), [na] "+v"(na), [nb] "=&v"(nb), [t] "+v"(tsh), [smk] "+v"(smk), [pmin] "+v"(pmin),
                                       [addr] "+v"(addr), [sm] "=&v"(t_sm), [lz] "=&v"(t_lz), [a4] "=&v"(t_a4)
                                     : [row] "v"(rowa), [kp1] "v"(kp1)
                                     : "vcc", "v248", "v249", "v250", "v251", "memory");
#undef FG_RG4
#undef FG_RC
                        const uint32_t sm = tsh & 31;
                        const bool toolong = pmin < (1u << (kp1 - 1));          // some code: leading zeros + k + 1 > 32 (or an all-zero window)
                        const uint32_t w2 = na, w3 = *(const uint32_t *)((const char *)br.ring + ((br.wb - 4 + (addr - a0)) & (FG_RG * 16 - 1))),
                                       wb = br.wb - 4 + (addr - a0);
#undef FG_WST
                        // a code longer than the window (or an all-zero window) voids the attempt
                        if (__any(toolong)) br.restore(keep);
                        else { br.w0 = w0; br.w1 = w1; br.w2 = w2; br.w3 = w3; br.wb = wb + 4; br.s = sm; }
                        if (!__any(toolong)) { tk = k; ii = FG_TS; tp[5]++; } else tp[6]++;
                    }
                    if (ii < FG_TS) tp[7]++;
                    for (; ii < FG_TS; ii++) {
                        const uint32_t i = i0 + ii;
                        if (i < start || i >= n) continue;
                        int32_t v;
                        if (kind == 2) v = br.sbits(sb);
                        else {
                            while (i >= pend) {
                                k = br.bits(plen);
                                is_esc = (k == escv);
                                if (is_esc) raw = br.bits(5);
                                if (rparams && part < FG_DEC_RPARAMS) rparams[((size_t)f * C + ch) * FG_DEC_RPARAMS + part] = (uint16_t)(is_esc ? (0x8000u | (raw << 8)) : k);
                                part++;
                                pend = po == 0 ? n : part * psz;
                            }
                            if (is_esc) v = br.sbits(raw);
                            else {
                                const uint32_t p = br.peek();
                                const uint32_t lz = (uint32_t)__clz(p);
                                uint32_t u;
                                if (lz + 1 + k <= 32) {
                                    const uint32_t rest = (p << lz) << 1;
                                    u = (lz << k) | (k ? (rest >> (32 - k)) : 0);
                                    br.consume(lz + 1 + k);
                                }
                                else {
                                    const uint32_t msb = br.unary(end_bits);
                                    u = (msb << k) | br.bits(k);
                                }
                                v = unzig(u);
                            }
                        }
                        row[ii] = (uint32_t)v;
                        tmask |= (u64)1 << ii;
                    }
                }
            }
            FG_TICK(3);
            if (isconst) tk = 0xFE;
            // ---- hand the tile to the helper wave: the row's facts, then the barrier (it also orders the LDS writes)
            {
                uint32_t *m = metas + (it & 1) * 64 * FG_META + lane * FG_META;
                // (rn << 8) | tk: block sizes are below 2^16, rn = 0 for idle lanes; output offset; mask; constant value -- one
                // 16-byte and one 8-byte store (the rows are 32 bytes apart)
                *(uint4 *)m = make_uint4((rn << 8) | tk, (uint32_t)roff, (uint32_t)(roff >> 32), (uint32_t)tmask);
                *(uint2 *)(m + 4) = make_uint2((uint32_t)(tmask >> 32), cval);
            }
            if (FUSED) {
                *(uint2 *)(L.feed + 2 * lane) = make_uint2(br.wb, br.H);          // read position and what this lane fetched itself, for the feeding wave
                cg_e2 = cg_e1; cg_e1 = br.wb >> 4;
            }
            __syncthreads();
            it++;
            if (alive && br.over) { err = 3; alive = false; rn = 0; }      // (see hlim: the generic decoder takes the frame)
            else if (act && br.pos() > end_bits) { err = 4; alive = false; rn = 0; }
            FG_TICK(4);
        }
    }
    if (accepted) {
        if (br.over) err = 3;              // (only ever set while the lane was still parsing: what followed is not to be trusted)
        if (!err) {
            const uint32_t endb = (br.pos() + 7) & ~7u;
            const uint32_t padb = endb - br.pos();
            if (endb != end_bits) err = 4;
            if (padb && (br.peek() >> (32 - padb)) != 0) err = 5;      // libFLAC read_zero_padding_: lost sync
        }
        results[f].err = err;
    }
    if (FUSED) { __syncthreads(); __syncthreads(); }          // the recurrence and the output waves finish the last two tiles
    if (FUSED && prof) {
        // [6]: distinct SIMDs under the four waves, [7]: 1 when another wave sits on the parser's SIMD
        const uint32_t a = ctrl[4], b = ctrl[5], c = ctrl[6], d = ctrl[7];
        tp[6] = ((1u << a) | (1u << b) | (1u << c) | (1u << d)) == 15u ? 4 : __popc((1u << a) | (1u << b) | (1u << c) | (1u << d));
        tp[7] = (a == b || a == c || a == d) ? 1 : 0;
    }
    if (prof && lane == 0) for (int i = 0; i < 8; i++) prof[(size_t)blockIdx.x * 8 + i] = tp[i];
#undef FG_TICK
#undef tile
}

__global__ void __launch_bounds__(128)
fg_dec_rice_kernel(const uint8_t *stream, u64 stream_len, const FgDecFrame *frames, uint32_t nframes, uint32_t G, uint32_t narrow,
                   int32_t *scratch, FgDecSub *subs, FgDecResult *results, u64 *prof, uint16_t *rparams)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dsm[];
    FgParseLds L;
    L.rings = dsm;                               // G rows of FG_RSTR words
    L.tiles = dsm + G * FG_RSTR;                 // two buffers of G rows of FG_TSTR words
    L.metas = L.tiles + 2 * G * FG_TSTR;         // two buffers of 64 rows of FG_META words
    L.ctrl = L.metas + 2 * 64 * FG_META;         // [0] tiles per launch group, [1] tiles per channel
    L.subp = nullptr; L.frm = nullptr; L.feed = nullptr;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x >= 64) {
        __syncthreads();
        const uint32_t T = L.ctrl[0], tpc = L.ctrl[1];
        for (uint32_t it = 0; it < T; it++) {
            __syncthreads();
            fg_dec_flush_tile(L.tiles + (it & 1) * G * FG_TSTR, L.metas + (it & 1) * 64 * FG_META, G, (it % tpc) * FG_TS, lane, scratch);
        }
        return;
    }
    fg_parse_wave<false>(stream, stream_len, frames, nframes, G, narrow, subs, results, prof, rparams, L, lane);
}

// CRC-16 (poly 0x8005, init 0) of frame bytes [0, bytes-2), compared with the stored big-endian CRC.
//
// No tables (round 3): the polynomial x^16 + x^15 + x^2 + 1 is sparse enough for a closed form -- for a 16-bit u,
//     u * x^16 mod P = (u << 1 ^ u << 2) & 0xFFFF ^ parity(u) * 0x8003 ^ u[15] * 0x000A ^ u[14] * 0x8005
// (checked over all 65536 values), so a 32-bit word w takes the state c to S(S(c ^ w >> 16) ^ (w & 0xFFFF)) in ~30 VALU
// instructions and no LDS look-up; the table version did six look-ups per word, at random addresses, beside a parser whose
// walks live on LDS.  Lanes own interleaved 16-byte granules (one unaligned 16-byte load a step): state * x^8192 + crc(granule),
// folded at the end with x^(128 (63 - lane)) (one multiplication by a constant out of a table).  Round 4: the per-step factor
// x^8192 has a closed form of its own (crc16_mul_x8192), and the remainder of the bytes in front of the first aligned word rides
// along as a granule in front of granule 0 instead of being raised to x^(128 G) by repeated squaring -- 1200 instructions a frame.
#endif  // FG_LEGACY
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
        const u64 o0 = offsets[f], o1 = offsets[f + 1];
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


#ifdef FG_LEGACY        // round 1 restore kernel and round 2 fused decoder
// ------------------------------------------------------------------------------------------------ restore
#define FG_TR 192           // samples per tile and chain: a multiple of both history lengths (8 and 12) and of the pass width
#define FG_TRS 196          // LDS row stride in words (16-byte aligned rows, rows of neighbouring lanes on different banks)
#define FG_TP 64            // columns moved per I/O pass (lane = column)
#define FG_RROWS 32         // chains per wave at most (one prefetch register per row)

// MAXO steps of s[i] = r[i] + ((sum_j q[j] * s[i-1-j]) >> shift).  Sample i lives in history slot i mod MAXO, so every
// register index below is a compile-time constant.  GATE: the group may still contain warm-up samples (i < order).
template <int MAXO, bool WIDE, bool GATE>
__device__ __forceinline__ void restore_group(int32_t (&h)[FG_DMAXO], const int32_t (&q)[FG_DMAXO], int shift, uint32_t order,
                                              uint32_t ibase, uint32_t *rowp)
{
    int32_t r[MAXO];
#pragma unroll
    for (int u = 0; u < MAXO; u += 4) {
        const uint4 t = *(const uint4 *)(rowp + u);
        r[u] = (int32_t)t.x; r[u + 1] = (int32_t)t.y; r[u + 2] = (int32_t)t.z; r[u + 3] = (int32_t)t.w;
    }
#pragma unroll
    for (int u = 0; u < MAXO; u++) {
        int32_t pred;
        if (!WIDE) {
            int32_t sum = 0;
#pragma unroll
            for (int j = MAXO - 1; j >= 0; j--) sum += __mul24(q[j], h[(u - 1 - j + 2 * MAXO) % MAXO]);
            pred = sum >> shift;
        }
        else {
            i64 sum = 0;
#pragma unroll
            for (int j = MAXO - 1; j >= 0; j--) sum += (i64)q[j] * (i64)h[(u - 1 - j + 2 * MAXO) % MAXO];
            pred = (int32_t)(sum >> shift);
        }
        int32_t v = r[u] + pred;
        if (GATE) v = (ibase + (uint32_t)u >= order) ? v : r[u];
        h[u] = v;
        r[u] = v;
    }
#pragma unroll
    for (int u = 0; u < MAXO; u += 4) *(uint4 *)(rowp + u) = make_uint4((uint32_t)r[u], (uint32_t)r[u + 1], (uint32_t)r[u + 2], (uint32_t)r[u + 3]);
}

// groups [g0, g1) of the current tile
template <int MAXO, bool WIDE>
__device__ __forceinline__ void restore_range(int32_t (&h)[FG_DMAXO], const int32_t (&q)[FG_DMAXO], int shift, uint32_t order,
                                              bool first_tile, uint32_t g0, uint32_t g1, uint32_t *rowp)
{
    for (uint32_t g = g0; g < g1; g++) {
        if (first_tile && g * MAXO < FG_DMAXO) restore_group<MAXO, WIDE, true>(h, q, shift, order, g * MAXO, rowp + g * MAXO);
        else restore_group<MAXO, WIDE, false>(h, q, shift, order, 0, rowp + g * MAXO);
    }
}

// lane = chain = (frame, channel); G frames per wave, G * C <= FG_RROWS.
//
// A tile is moved in three passes of 64 columns.  Per pass the wave loads one register per chain (coalesced, the row's
// base address and length come from the owning lane with v_readlane), parks it in LDS one phase later, and the phase in
// between restores a third of the previous data -- HBM latency hides behind the recurrence:
//   phase 0: land(t,0)  issue(t,1)    restore groups of columns   0.. 63
//   phase 1: land(t,1)  issue(t,2)    restore groups of columns  64..127   write out pass 0
//   phase 2: land(t,2)  issue(t+1,0)  restore the rest                     write out passes 1, 2
// (group boundaries of the 12-tap variant do not fall on pass boundaries; each phase restores the groups that are
// complete, which is why pass p is written out one phase later.)
template <bool WIDE>
__global__ void __launch_bounds__(128)
fg_dec_restore_kernel(const FgDecFrame *frames, uint32_t nframes, uint32_t G, uint32_t C, const FgDecSub *subs, const int32_t *scratch,
                      int32_t *out, FgDecResult *results, uint32_t interleave, u64 *prof)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];     // rows of FG_TRS words: chains (rounded up to 16) + 1 spare
    // Two waves per workgroup with the same lane -> (frame, channel) mapping: wave 0 runs the serial recurrence on the LDS
    // tile, wave 1 moves the data (scratch -> tile ahead of it, tile -> output behind it).  One barrier per 64-column step.
    const int lane = threadIdx.x & 63;
    const bool mover = threadIdx.x >= 64;
    const uint32_t chains = G * C;
    const uint32_t fi = (uint32_t)lane / C, ch = (uint32_t)lane % C;
    const uint32_t f = blockIdx.x * G + fi;
    const bool mine = (uint32_t)lane < chains && f < nframes;
    uint32_t n = 0, status = 1, ca = 0, crcw = 0;
    u64 out_off = 0;
    if (mine) {
        const FgDecFrame fr = frames[f];
        if (fr.bytes != 0 && fr.channels == C) {
            n = fr.n; status = results[f].err; ca = fr.ca; out_off = fr.out_off;
            const uint32_t cw = results[f].crc;
            if (status == 0 && (cw & 0x80000000u)) status = 2;          // CRC-16 mismatch (fg_dec_crc_kernel)
            crcw = cw;
        }
    }
    __syncthreads();                                                    // both waves have read the parse / CRC results
    if (!mover && mine && n != 0 && ch == 0) { results[f].err = status; results[f].crc = crcw & 0xFFFFu; }
    const bool ok = mine && n != 0 && status == 0;
    int32_t q[FG_DMAXO], h[FG_DMAXO];
#pragma unroll
    for (int j = 0; j < FG_DMAXO; j++) { q[j] = 0; h[j] = 0; }
    uint32_t order = 0, wasted = 0;
    int shift = 0;
    if (ok) {
        const FgDecSub *sd = &subs[(size_t)f * C + ch];
        order = sd->order; shift = sd->shift; wasted = sd->wasted;
#pragma unroll
        for (int j = 0; j < FG_DMAXO; j++) if ((uint32_t)j < order) q[j] = sd->q[j];
    }
    // row facts, read by the I/O passes with v_readlane
    const uint32_t n_in = ok ? n : 0;                                   // residuals to load
    const uint32_t n_out = (mine && status != 3) ? n : 0;               // samples to write (status 3: the generic kernel writes)
    const u64 plane = ok ? out_off * C + (u64)ch * n : 0;               // this chain's residual plane in `scratch`
    const uint32_t nmax = wave_max32(n_out);
    const bool big = __any(ok && order > 8);
    const bool scratch_aligned = (((uintptr_t)scratch) & 15) == 0, out_aligned = (((uintptr_t)out) & 15) == 0;
    // idle lanes run the recurrence on a spare row behind the real ones
    uint32_t *rowp = &tile[((uint32_t)lane < chains ? (uint32_t)lane : ((chains + 15) & ~15u)) * FG_TRS];
    u64 tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = prof ? clock64() : 0;
#define FG_TICK(i) do { if (prof) { const u64 now_ = clock64(); tp[i] += now_ - tlast; tlast = now_; } } while (0)

    // ---- tile I/O.  Fast form (whole 64-column pass inside every row, 16-byte aligned planes): lane = (row, quarter) for
    // the residuals in (four 16-byte loads per lane and round of 16 rows), lane = (frame, eighth) for the samples out
    // (16-byte LDS reads and stores); the row / frame facts sit in per-lane registers, fetched once by permute.
    // General form (tails, odd block sizes, other channel counts): one row per step, lane = column, facts by v_readlane.
    const uint32_t nrnd = (chains + 15) >> 4;                            // rounds of 16 rows (1 or 2)
    uint32_t l_rn[2], f_rn[2], f_ok[2], f_ca[2], f_wa[2], f_wb[2];
    u64 l_base[2], f_oo[2];
    bool planes_aligned = true;
#pragma unroll
    for (int R = 0; R < 2; R++) {
        const int row = R * 16 + (lane >> 2);
        l_rn[R] = (uint32_t)__shfl((int)n_in, row);
        l_base[R] = ((u64)(uint32_t)__shfl((int)(uint32_t)(plane >> 32), row) << 32) | (uint32_t)__shfl((int)(uint32_t)plane, row);
        const int fl = 2 * (R * 8 + (lane >> 3));                        // lane of the frame's first chain (stereo)
        f_rn[R] = (uint32_t)__shfl((int)n_out, fl & 63);
        f_ok[R] = (uint32_t)__shfl((int)n_in, fl & 63);
        f_ca[R] = (uint32_t)__shfl((int)ca, fl & 63);
        f_wa[R] = (uint32_t)__shfl((int)wasted, fl & 63);
        f_wb[R] = (uint32_t)__shfl((int)wasted, (fl + 1) & 63);
        f_oo[R] = ((u64)(uint32_t)__shfl((int)(uint32_t)(out_off >> 32), fl & 63) << 32) | (uint32_t)__shfl((int)(uint32_t)out_off, fl & 63);
        if (R * 8 + (lane >> 3) >= (int)G) { f_rn[R] = 0; f_ok[R] = 0; }
    }
    planes_aligned = !__any(((plane & 3) != 0 && n_in != 0) || ((out_off & 1) != 0 && n_out != 0) || ((n & 3) != 0 && n_out != 0));
    const uint32_t nmin_in = ~wave_max32(~(n_in ? n_in : 0xFFFFFFFFu));   // shortest row that is loaded at all
    const uint32_t nmin_out = ~wave_max32(~(n_out ? n_out : 0xFFFFFFFFu));
    fg_u32x4 pfv[2][4];
    uint32_t pf[FG_RROWS];
    bool pfvec = false;
    const uint32_t nr8 = (chains + 7) >> 3;
    auto issue8 = [&](int r0, uint32_t i) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t rn = rl(n_in, r0 + r);
            const u64 base = ((u64)rl((uint32_t)(plane >> 32), r0 + r) << 32) | rl((uint32_t)plane, r0 + r);
            pf[r0 + r] = (uint32_t)scratch[base + (i < rn ? i : 0)];
        }
    };
    auto issue = [&](uint32_t i0, uint32_t p) __attribute__((always_inline)) {
        const uint32_t c0 = i0 + p * FG_TP;
        pfvec = planes_aligned && scratch_aligned && c0 + FG_TP <= nmin_in;
        if (pfvec) {
            const uint32_t i = c0 + ((uint32_t)lane & 3) * 16;
#pragma unroll
            for (int R = 0; R < 2; R++) {
                if ((uint32_t)R < nrnd) {
                    const fg_u32x4 *src = (const fg_u32x4 *)(scratch + l_base[R] + (l_rn[R] ? i : 0));
#pragma unroll
                    for (int t = 0; t < 4; t++) pfv[R][t] = src[t];
                }
            }
        }
        else {
            const uint32_t i = c0 + (uint32_t)lane;
            issue8(0, i);
            if (nr8 > 1) issue8(8, i);
            if (nr8 > 2) issue8(16, i);
            if (nr8 > 3) issue8(24, i);
        }
    };
    auto land8 = [&](int r0, uint32_t p) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; r++) tile[(r0 + r) * FG_TRS + p * FG_TP + lane] = pf[r0 + r];
    };
    auto land = [&](uint32_t p) __attribute__((always_inline)) {
        if (pfvec) {
#pragma unroll
            for (int R = 0; R < 2; R++) {
                if ((uint32_t)R < nrnd) {
                    fg_u32x4 *dst = (fg_u32x4 *)&tile[(R * 16 + (lane >> 2)) * FG_TRS + p * FG_TP + (lane & 3) * 16];
#pragma unroll
                    for (int t = 0; t < 4; t++) dst[t] = pfv[R][t];
                }
            }
        }
        else {
            land8(0, p);
            if (nr8 > 1) land8(8, p);
            if (nr8 > 2) land8(16, p);
            if (nr8 > 3) land8(24, p);
        }
        wave_lds_fence();
    };
    auto writeout = [&](uint32_t i0, uint32_t p) __attribute__((always_inline)) {
        const uint32_t c0 = i0 + p * FG_TP;
        if (C == 2 && planes_aligned && out_aligned && c0 + FG_TP <= nmin_out) {
            const uint32_t colb = p * FG_TP + ((uint32_t)lane & 7) * 8;
#pragma unroll
            for (int R = 0; R < 2; R++) {
                if ((uint32_t)(R * 8) < G && f_rn[R] != 0) {
                    const uint32_t r0 = 2 * (R * 8 + ((uint32_t)lane >> 3));
                    const uint32_t i = i0 + colb;
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        int32_t a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
                        if (f_ok[R]) {
                            const uint4 ta = *(const uint4 *)&tile[r0 * FG_TRS + colb + 4 * hh];
                            const uint4 tb = *(const uint4 *)&tile[(r0 + 1) * FG_TRS + colb + 4 * hh];
                            const uint32_t xa[4] = {ta.x, ta.y, ta.z, ta.w}, xb[4] = {tb.x, tb.y, tb.z, tb.w};
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                int32_t av = (int32_t)(xa[e] << f_wa[R]), bv = (int32_t)(xb[e] << f_wb[R]);
                                const uint32_t cc = f_ca[R];
                                // (32-bit streams: a side channel with wasted bits is a 33-bit value once shifted back)
                                const i64 side = WIDE ? (i64)((u64)(i64)(int32_t)xb[e] << f_wb[R]) : (i64)bv;
                                const i64 mid = (i64)(((u64)(i64)av) << 1) | (side & 1);
                                const int32_t ma = (int32_t)((mid + side) >> 1), mb = (int32_t)((mid - side) >> 1);
                                a[e] = cc == 2 ? av + bv : cc == 3 ? ma : av;
                                b[e] = cc == 1 ? av - bv : cc == 3 ? mb : bv;
                            }
                        }
                        int32_t *o = out + f_oo[R] * 2;
                        if (interleave) {
                            int4 *d = (int4 *)(o + (size_t)(i + 4 * hh) * 2);
                            d[0] = make_int4(a[0], b[0], a[1], b[1]);
                            d[1] = make_int4(a[2], b[2], a[3], b[3]);
                        }
                        else {
                            *(int4 *)(o + i + 4 * hh) = make_int4(a[0], a[1], a[2], a[3]);
                            *(int4 *)(o + f_rn[R] + i + 4 * hh) = make_int4(b[0], b[1], b[2], b[3]);
                        }
                    }
                }
            }
            return;
        }
        const uint32_t col = p * FG_TP + (uint32_t)lane, i = i0 + col;
        if (C == 2) {
            for (uint32_t g = 0; g < G; g++) {
                const int r0 = (int)(2 * g);
                const uint32_t rn = rl(n_out, r0);
                if (i0 + p * FG_TP >= rn) continue;
                const uint32_t okr = rl(n_in, r0), cc = rl(ca, r0), wa = rl(wasted, r0), wb = rl(wasted, r0 + 1);
                const u64 oo = ((u64)rl((uint32_t)(out_off >> 32), r0) << 32) | rl((uint32_t)out_off, r0);
                int32_t a = 0, b = 0;
                if (okr) {
                    a = (int32_t)(tile[r0 * FG_TRS + col] << wa);
                    b = (int32_t)(tile[(r0 + 1) * FG_TRS + col] << wb);
                    if (cc == 1) b = a - b;
                    else if (cc == 2) a = a + b;
                    else if (cc == 3) {
                        const i64 side = WIDE ? (i64)((u64)(i64)(int32_t)tile[(r0 + 1) * FG_TRS + col] << wb) : (i64)b;
                        const i64 mid = (i64)(((u64)(i64)a) << 1) | (side & 1);
                        a = (int32_t)((mid + side) >> 1);
                        b = (int32_t)((mid - side) >> 1);
                    }
                }
                if (i < rn) {
                    int32_t *o = out + oo * 2;
                    if (interleave) ((int2 *)o)[i] = make_int2(a, b);
                    else { o[i] = a; o[rn + i] = b; }
                }
            }
        }
        else {
            for (uint32_t r = 0; r < chains; r++) {
                const uint32_t rn = rl(n_out, (int)r);
                if (i0 + p * FG_TP >= rn) continue;
                const uint32_t okr = rl(n_in, (int)r), wa = rl(wasted, (int)r);
                const u64 oo = ((u64)rl((uint32_t)(out_off >> 32), (int)r) << 32) | rl((uint32_t)out_off, (int)r);
                const int32_t v = okr ? (int32_t)(tile[r * FG_TRS + col] << wa) : 0;
                if (i < rn) {
                    int32_t *o = out + oo * C;
                    if (interleave) o[(size_t)i * C + (r % C)] = v;
                    else o[(size_t)(r % C) * rn + i] = v;
                }
            }
        }
    };
    const uint32_t ng = big ? FG_TR / 12 : FG_TR / 8;
    const uint32_t ga = big ? 5 : 8, gb = big ? 10 : 16;       // groups that are complete after passes 0 and 1 have landed

    // Steps s = 3 * tile + pass.  Restore step (t, p) covers the sample groups of columns [64p, 64p + 64) (order > 8: 12-sample
    // groups, so the first two steps end at columns 60 and 120; block 0 is complete after step (t, 1), blocks 1 and 2 after
    // (t, 2)).  The mover stays off the block the recurrence is working on:
    //   step (t, 0): write out (t-1, 1); land (t, 1); write out (t-1, 2); request (t, 2)
    //   step (t, 1): land (t, 2); request (t+1, 0)
    //   step (t, 2): write out (t, 0); land (t+1, 0); request (t+1, 1)
    const uint32_t T = (nmax + FG_TR - 1) / FG_TR, S = 3 * T;
    if (mover) {
        if (T) { issue(0, 0); land(0); issue(0, 1); }
        for (uint32_t s = 0; s <= S; s++) {
            __syncthreads();
            const uint32_t t = s / 3, p = s - 3 * t, i0 = t * FG_TR;
            if (p == 0) {
                if (t > 0) writeout(i0 - FG_TR, 1);
                if (t < T) land(1);
                if (t > 0) writeout(i0 - FG_TR, 2);
                if (t < T) issue(i0, 2);
            }
            else if (p == 1) {
                land(2);
                if (t + 1 < T) issue(i0 + FG_TR, 0);
            }
            else {
                writeout(i0, 0);
                if (t + 1 < T) { land(0); issue(i0 + FG_TR, 1); }
            }
        }
        return;
    }
    for (uint32_t i0 = 0; i0 < nmax; i0 += FG_TR) {
        const bool first = i0 == 0;
        __syncthreads();
        FG_TICK(0);
        if (big) restore_range<12, WIDE>(h, q, shift, order, first, 0, ga, rowp); else restore_range<8, WIDE>(h, q, shift, order, first, 0, ga, rowp);
        __syncthreads();
        FG_TICK(2);
        if (big) restore_range<12, WIDE>(h, q, shift, order, first, ga, gb, rowp); else restore_range<8, WIDE>(h, q, shift, order, first, ga, gb, rowp);
        __syncthreads();
        FG_TICK(2);
        if (big) restore_range<12, WIDE>(h, q, shift, order, first, gb, ng, rowp); else restore_range<8, WIDE>(h, q, shift, order, first, gb, ng, rowp);
        FG_TICK(2);
    }
    __syncthreads();
    if (prof && lane == 0) for (int i = 0; i < 8; i++) prof[(size_t)blockIdx.x * 8 + i] = tp[i];
#undef FG_TICK
}

// ------------------------------------------------------------------------------------------------ fused decode
// One workgroup of four waves per group of G frames; the tiles of 64 samples per frame travel through LDS:
//   wave 0  parser       lane = frame    bit-serial: delimits the Rice codes of tile t            (fg_parse_wave)
//   wave 1  converter    16 lanes / row  windows of tile t-1 -> residuals, into a ring of three residual tiles in LDS
//   wave 2  recurrence   lane = frame    s[i] = r[i] + (sum q*s >> shift) over tile t-2, history in registers
//   wave 3  output       lane = column   tile t-3: channel 0 of a stereo frame is parked in HBM (one int32 plane per frame),
//                                        channel 1 fetches it back, undoes the decorrelation and stores both, interleaved
// One barrier per tile.  The recurrence (78 cycles a sample) is faster than the parse (104 cycles a code), so the kernel takes
// what the parse takes; the residual plane of the two-kernel version (4 bytes per sample out and in again) is gone, what
// remains is the channel-0 plane.  Frames the kernels cannot take (status 3) and frames that fail are settled afterwards
// (generic kernel / fg_dec_fix_kernel writes silence).
#define FG_RT 3             // residual tiles in flight
#define FG_FUSED_GMAX 48    // frames per workgroup of the fused kernel at most (150 KB of LDS)

template <int MAXO, bool WIDE, bool GATE>
__device__ __forceinline__ void frestore_group(int32_t (&h)[16], const int32_t (&q)[16], int shift, uint32_t order, uint32_t ibase, uint32_t *rowp)
{
    int32_t r[MAXO];
#pragma unroll
    for (int u = 0; u < MAXO; u += 4) {
        const uint4 t = *(const uint4 *)(rowp + u);
        r[u] = (int32_t)t.x; r[u + 1] = (int32_t)t.y; r[u + 2] = (int32_t)t.z; r[u + 3] = (int32_t)t.w;
    }
#pragma unroll
    for (int u = 0; u < MAXO; u++) {
        int32_t pred;
        if (!WIDE) {
            int32_t sum = 0;
#pragma unroll
            for (int j = (MAXO == 8 ? 7 : 11); j >= 0; j--) sum += __mul24(q[j], h[(u - 1 - j + 2 * MAXO) % MAXO]);
            pred = sum >> shift;
        }
        else {
            i64 sum = 0;
#pragma unroll
            for (int j = (MAXO == 8 ? 7 : 11); j >= 0; j--) sum += (i64)q[j] * (i64)h[(u - 1 - j + 2 * MAXO) % MAXO];
            pred = (int32_t)(sum >> shift);
        }
        int32_t v = r[u] + pred;
        if (GATE) v = (ibase + (uint32_t)u >= order) ? v : r[u];
        h[u] = v;
        r[u] = v;
    }
#pragma unroll
    for (int u = 0; u < MAXO; u += 4) *(uint4 *)(rowp + u) = make_uint4((uint32_t)r[u], (uint32_t)r[u + 1], (uint32_t)r[u + 2], (uint32_t)r[u + 3]);
}

// converter: tile of windows -> tile of residuals in LDS (the LDS twin of fg_dec_flush_tile); rn[row] = samples of the row
__device__ __forceinline__ void fg_dec_convert_tile(const uint32_t *tile, const uint32_t *meta, uint32_t G, uint32_t i0, int lane, uint32_t *dst,
                                                    uint32_t *rn_out, int32_t *warm, const uint32_t *frm, uint32_t fbase, uint32_t ch)
{
    if ((uint32_t)lane < G) rn_out[lane] = meta[lane * FG_META] >> 8;
    uint32_t special = 0;
    if ((uint32_t)lane < G) {
        const uint32_t m = meta[lane * FG_META], rn_l = m >> 8, tk_l = m & 0xFF;
        special = rn_l > i0 && (tk_l >= 0xFE || (meta[lane * FG_META + 3] | meta[lane * FG_META + 4]) != 0);
    }
    if (!__any(special)) {
        const uint32_t rsub = (uint32_t)lane >> 4, q4 = ((uint32_t)lane & 15) * 4;
        for (uint32_t r0 = 0; r0 < G; r0 += 4) {
            const uint32_t r = r0 + rsub;
            const uint32_t src = r < G ? r : 0;
            const uint32_t m = meta[src * FG_META];
            const uint32_t rn_r = r < G ? (m >> 8) : 0, kk = m & 0xFF;
            if (i0 + q4 < rn_r) {
                const uint4 pw = *(const uint4 *)&tile[r * FG_TSTR + q4];
                const uint32_t p4[4] = {pw.x, pw.y, pw.z, pw.w};
                uint32_t res[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t lz = (uint32_t)__clz(p4[e]);
                    const uint32_t rest = (p4[e] << lz) << 1;
                    const uint32_t u = (lz << kk) | (kk ? (rest >> (32 - kk)) : 0);
                    res[e] = (uint32_t)unzig(u);
                }
                *(uint4 *)&dst[r * FG_TSTR + q4] = make_uint4(res[0], res[1], res[2], res[3]);
            }
        }
        return;
    }
    for (uint32_t r = 0; r < G; r++) {
        const uint32_t m = meta[r * FG_META], rn_s = m >> 8, kk = m & 0xFF;
        if (i0 >= rn_s) continue;
        uint32_t val;
        if (kk == 0xFE) val = meta[r * FG_META + 5];
        else {
            val = tile[r * FG_TSTR + lane];
            if (kk != 0xFF) {
                const uint32_t mlo = meta[r * FG_META + 3], mhi = meta[r * FG_META + 4];
                const uint32_t done = ((lane < 32 ? mlo : mhi) >> (lane & 31)) & 1;
                const uint32_t lz = (uint32_t)__clz(val);
                const uint32_t rest = (val << lz) << 1;
                const uint32_t u = (lz << kk) | (kk ? (rest >> (32 - kk)) : 0);
                val = done ? val : (uint32_t)unzig(u);
            }
        }
        dst[r * FG_TSTR + lane] = val;
        // FLAC__Frame.subframes[].warmup: the first samples of the subframe as coded (they sit in the first tile)
        if (warm && i0 == 0 && lane < 32) {
            const uint32_t C = frm[r * FG_FRM + 1];
            if (ch < C) warm[((size_t)(fbase + r) * C + ch) * 32 + lane] = (int32_t)val;
        }
    }
}

// NR: rounds of 16 rows the output wave makes per tile (2 up to 32 frames per workgroup, 3 up to 48)
template <bool WIDE, int NR>
__global__ void __launch_bounds__(256)
fg_dec_fused_kernel(const uint8_t *stream, u64 stream_len, const FgDecFrame *frames, uint32_t nframes, uint32_t G, uint32_t narrow,
                    int32_t *scratch, FgDecSub *subs, FgDecResult *results, uint16_t *rparams, int32_t *warm, int32_t *out,
                    uint32_t interleave, u64 *prof)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t dsm[];
    FgParseLds L;
    L.rings = dsm;
    L.tiles = dsm + G * FG_RSTR;
    L.metas = L.tiles + 2 * G * FG_TSTR;
    L.ctrl = L.metas + 2 * 64 * FG_META;
    uint32_t *const rt = L.ctrl + 8;                          // FG_RT residual tiles of (G + 1) rows (the last one is a spare)
    uint32_t *const rnm = rt + FG_RT * (G + 1) * FG_TSTR;     // samples per row, per residual tile: FG_RT x 64
    L.subp = rnm + FG_RT * 64;                                // 4 x G x FG_SUBP
    L.frm = L.subp + 4 * G * FG_SUBP;                         // G x FG_FRM
    L.feed = L.frm + G * FG_FRM;                              // [0, 128): the parser's (read position, own fill level) pairs, [128, 192): fill levels from the feeder
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    if (prof && lane == 0) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        L.ctrl[4 + wave] = (hwid >> 4) & 3;          // SIMD of this wave (FLACGPU_DEC_PROF=2 reports how many the four share)
    }
    if (wave == 0) {
        fg_parse_wave<true>(stream, stream_len, frames, nframes, G, narrow, subs, results, prof, rparams, L, lane);
        return;
    }
    // FLACGPU_DEC_PROF=2: clock64() ticks every helper wave spends waiting at the tile barrier, and its total (tuning aid)
    u64 pw_wait = 0, pw_t0 = prof ? clock64() : 0;
#define FG_BAR() do { if (prof) { const u64 a_ = clock64(); __syncthreads(); pw_wait += clock64() - a_; } else __syncthreads(); } while (0)
#define FG_PROF_END() do { if (prof && lane == 0) { u64 *pp_ = prof + ((size_t)((nframes + 63) & ~63u) + blockIdx.x) * 8 + wave * 2; pp_[0] = pw_wait; pp_[1] = clock64() - pw_t0; } } while (0)
    FG_BAR();
    const uint32_t T = L.ctrl[0], tpc = L.ctrl[1];
    const uint32_t fbase = blockIdx.x * G;
    if (wave == 1) {
        // ---- converter, and feeder of the parser's rings (lane = frame): the parser spends a third of its time per tile
        // requesting, byte-swapping and parking the eight groups its lanes consume -- this wave has that time to spare.  Per
        // tile: park what was requested a tile ago, publish the fill level (L.feed[128 + lane]; the parser reads it after
        // the next barrier), request the next groups behind the parser's read position as it was at the end of the last
        // tile (L.feed[2 lane]).  Three tiles pass between a read position and the data it asked for being usable, so the
        // rings start 40 groups full (five round trips before the parser's first word) and the parser keeps its own
        // fetch for lanes that run short.  Parked slots lie beyond everything the parser may read (same bound as before,
        // taken from an older -- smaller -- read position), and a group both waves happen to fetch has the same bytes.
        BitRd fd;
        fd.fg = nullptr; fd.glim = 0; fd.skip0 = 0; fd.ring = L.rings; fd.w0 = fd.w1 = fd.w2 = fd.w3 = 0; fd.s = 0; fd.wb = 12; fd.H = 0;
        fd.hlim = ~0u; fd.over = false;
        fd.pfH = 0; fd.pfn = 0; fd.pfvalid = false;
        bool feeding = false;
        {
            const uint32_t f = fbase + (uint32_t)lane;
            if ((uint32_t)lane < G && f < nframes) {
                const FgDecFrame fr = frames[f];
                if (fr.bytes != 0 && fr.bytes >= fr.hdr_bytes + 2) { feeding = true; fg_frame_reader(fd, stream, stream_len, fr, L.rings + lane * FG_RSTR); }
            }
            for (int r = 0; r < 5; r++) { fd.issue(feeding, false); fd.land(); }
            *(uint2 *)(L.feed + 2 * lane) = make_uint2(fd.wb, fd.H);
            L.feed[128 + lane] = fd.H;
        }
        FG_BAR();
        for (uint32_t j = 1; j <= T + 2; j++) {
            FG_BAR();
            if (j <= T) {
                const uint2 pp = *(const uint2 *)(L.feed + 2 * lane);          // the parser's read position and own fill level
                fd.wb = pp.x;
                fd.land();
                // (a lane that outran the feed fetched for itself: go on behind what it has, not behind our own count --
                // groups the parser has passed must not be parked over newer ones)
                fd.H = pp.y > fd.H ? pp.y : fd.H;
                L.feed[128 + lane] = fd.H;
                fd.issue(feeding, false);
            }
            if (j <= T) {
                const uint32_t it = j - 1;
                fg_dec_convert_tile(L.tiles + (it & 1) * G * FG_TSTR, L.metas + (it & 1) * 64 * FG_META, G, (it % tpc) * FG_TS, lane,
                                    rt + (it % FG_RT) * (G + 1) * FG_TSTR, rnm + (it % FG_RT) * 64, warm, L.frm, fbase, it / tpc);
            }
        }
        FG_PROF_END();
        return;
    }
    FG_BAR();                                     // (the rings' first fill, see the converter)
    if (wave == 2) {
        // ---- recurrence: after barrier j, tile j - 2.  lane = frame row; idle lanes work on the spare row.
        int32_t q[16], h[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { q[k] = 0; h[k] = 0; }
        uint32_t order = 0;
        int shift = 0;
        bool big = false;
        for (uint32_t j = 1; j <= T + 2; j++) {
            FG_BAR();
            if (j < 2 || j > T + 1 || (interleave & 0x200)) continue;
            const uint32_t it = j - 2, ch = it / tpc, i0 = (it % tpc) * FG_TS;
            if (i0 == 0) {
                // a new channel: this row's predictor
                order = 0; shift = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) { q[k] = 0; h[k] = 0; }
                if ((uint32_t)lane < G) {
                    const uint32_t *sp = L.subp + ((ch & 3) * G + lane) * FG_SUBP;
                    order = sp[0]; shift = (int)sp[1];
#pragma unroll
                    for (int k = 0; k < 12; k++) q[k] = (int32_t)sp[4 + k];
                }
                big = __any(order > 8);
            }
            uint32_t *rowp = rt + ((it % FG_RT) * (G + 1) + ((uint32_t)lane < G ? (uint32_t)lane : G)) * FG_TSTR;
            const bool first = i0 == 0;
            if (big) {
                // 12-tap history kept in 16 slots (64 = 4 x 16: the slot of a sample is a compile-time constant)
                frestore_group<16, WIDE, true>(h, q, shift, first ? order : 0, 0, rowp);
                frestore_group<16, WIDE, false>(h, q, shift, order, 0, rowp + 16);
                frestore_group<16, WIDE, false>(h, q, shift, order, 0, rowp + 32);
                frestore_group<16, WIDE, false>(h, q, shift, order, 0, rowp + 48);
            }
            else {
                frestore_group<8, WIDE, true>(h, q, shift, first ? order : 0, 0, rowp);
                frestore_group<8, WIDE, true>(h, q, shift, first ? order : 0, 8, rowp + 8);
#pragma unroll 1
                for (uint32_t g = 2; g < 8; g++) frestore_group<8, WIDE, false>(h, q, shift, order, 0, rowp + g * 8);
            }
        }
        FG_PROF_END();
        return;
    }
    // ---- output: after barrier j, tile j - 3.
    // Fast form (stereo, whole 64-sample tile inside every live row, 16-byte aligned planes): lane = (row, quarter): 16 samples
    // per lane and round of 16 rows, 16-byte LDS reads, loads and stores; the row's facts sit in the lane's registers (fetched
    // at the start of a channel).  Anything else (mono, more channels, tails, odd offsets): lane = column, row after row.
    const uint32_t nrnd = (G + 15) >> 4;
    uint32_t f_c[NR] = {}, f_ca[NR] = {}, f_n[NR] = {}, f_w[NR] = {};
    u64 f_oo[NR] = {};
    uint4 pa[NR][4];                               // channel 0 of the tile that is next for the fast form (prefetched), per round
    uint32_t pa_it[NR];               // ... and which tile that is
    // The parked channel is kept as 16-bit values while they fit (tile by tile, from the start of the frame: bytes
    // [128 k, 128 k + 128) of the frame's plane for tile k); from the first tile that holds a larger value -- the side
    // channel of a right-side frame, a predictor gone wild in a damaged or hand-made stream -- or that takes the general
    // form, the frame goes on in 32 bits at the usual place (bytes [256 k, ..): behind everything parked before).
    // f_wf = first 32-bit tile of the row's frame.
    uint32_t f_wf[NR];
#pragma unroll
    for (int R = 0; R < NR; R++) {
        pa_it[R] = ~0u;
#pragma unroll
        for (int t = 0; t < 4; t++) pa[R][t] = make_uint4(0, 0, 0, 0);
    }
    const bool out_al = (((uintptr_t)out) & 15) == 0 && (((uintptr_t)scratch) & 15) == 0;
    u64 pw_busy[2] = {0, 0}, pw_b0 = 0;
    uint32_t pw_ch = 0;
    for (uint32_t j = 1; j <= T + 2; j++) {
        if (prof && j > 3) pw_busy[pw_ch & 1] += clock64() - pw_b0;
        FG_BAR();
        if (prof) pw_b0 = clock64();
        if (j < 3 || (interleave & 0x100)) continue;
        const uint32_t it = j - 3, ch = it / tpc, i0 = (it % tpc) * FG_TS;
        pw_ch = ch;
        const uint32_t *tile = rt + (it % FG_RT) * (G + 1) * FG_TSTR;
        const uint32_t *rn_ = rnm + (it % FG_RT) * 64;
        if (i0 == 0) {
#pragma unroll
            for (int R = 0; R < NR; R++) {
                const uint32_t row = R * 16 + ((uint32_t)lane >> 2);
                const bool have = row < G;
                const uint32_t *fm = L.frm + (have ? row : 0) * FG_FRM;
                f_n[R] = have ? fm[0] : 0; f_c[R] = have ? fm[1] : 0; f_ca[R] = fm[2];
                f_oo[R] = ((u64)fm[4] << 32) | fm[3];
                f_w[R] = have ? L.subp[((ch & 3) * G + row) * FG_SUBP + 2] : 0;
                if (ch == 0) f_wf[R] = WIDE ? 0u : ~0u;
            }
        }
        const uint32_t tk = it % tpc;
        // is this tile one for the fast form?
        bool ok_fast = out_al && (ch < 2);
        {
            bool bad = false;
#pragma unroll
            for (int R = 0; R < NR; R++) {
                const uint32_t row = R * 16 + ((uint32_t)lane >> 2);
                if ((uint32_t)R < nrnd && row < G) {
                    const uint32_t rn = rn_[row];
                    if (rn > i0) bad |= (f_c[R] != 2) || (i0 + FG_TS > rn) || ((f_oo[R] & 1) != 0) || ((f_n[R] & 3) != 0);
                }
            }
            ok_fast = ok_fast && !__any(bad);
        }
        if (ok_fast) {
            const uint32_t cq = ((uint32_t)lane & 3) * 16;
            bool live[NR] = {};
#pragma unroll
            for (int R = 0; R < NR; R++) {
                const uint32_t row = R * 16 + ((uint32_t)lane >> 2);
                live[R] = (uint32_t)R < nrnd && row < G && rn_[row < G ? row : 0] > i0;
                // channel 0 of this stretch comes back from HBM: normally requested a tile ago (see the end of this block)
                if (live[R] && ch == 1 && pa_it[R] != it) {
                    const bool p16 = tk < f_wf[R];
                    const uint4 *src = (const uint4 *)((const char *)(scratch + f_oo[R] * 2) + (size_t)(i0 + cq) * (p16 ? 2 : 4));
                    pa[R][0] = src[0]; pa[R][1] = src[1];
                    if (!p16) { pa[R][2] = src[2]; pa[R][3] = src[3]; }
                }
            }
            // Every prefetched value is touched before the first store of the tile goes out: memory operations retire in
            // order, so a wait for one of these loads placed behind stores waits for the stores too (the compiler put a
            // full vmcnt(0) between the two rounds: two store round trips per tile).
            if (ch == 1) {
#pragma unroll
                for (int R = 0; R < NR; R++)
#pragma unroll
                    for (int t = 0; t < 4; t++) asm volatile("" : "+v"(pa[R][t].x), "+v"(pa[R][t].y), "+v"(pa[R][t].z), "+v"(pa[R][t].w));
            }
#pragma unroll
            for (int R = 0; R < NR; R++) {
                if (!live[R]) continue;
                const uint32_t row = R * 16 + ((uint32_t)lane >> 2);
                const uint4 *lt = (const uint4 *)&tile[row * FG_TSTR + cq];
                const uint32_t wsh = f_w[R];
                if (ch == 0) {
                    uint4 v[4];
                    uint32_t big = 0;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const uint4 x = lt[t];
                        v[t] = make_uint4(x.x << wsh, x.y << wsh, x.z << wsh, x.w << wsh);
                        big |= (v[t].x + 32768u) | (v[t].y + 32768u) | (v[t].z + 32768u) | (v[t].w + 32768u);
                    }
                    if (tk < f_wf[R]) {
                        // does the row's tile (this lane's quarter and its three neighbours') fit 16 bits?
                        big >>= 16;
                        big |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)big, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
                        big |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)big, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
                        if (big) f_wf[R] = tk;
                    }
                    char *plane = (char *)(scratch + f_oo[R] * 2);
                    if (tk < f_wf[R]) {
                        uint4 *dstp = (uint4 *)(plane + (size_t)(i0 + cq) * 2);
                        dstp[0] = make_uint4(__builtin_amdgcn_perm(v[0].y, v[0].x, 0x05040100u), __builtin_amdgcn_perm(v[0].w, v[0].z, 0x05040100u),
                                             __builtin_amdgcn_perm(v[1].y, v[1].x, 0x05040100u), __builtin_amdgcn_perm(v[1].w, v[1].z, 0x05040100u));
                        dstp[1] = make_uint4(__builtin_amdgcn_perm(v[2].y, v[2].x, 0x05040100u), __builtin_amdgcn_perm(v[2].w, v[2].z, 0x05040100u),
                                             __builtin_amdgcn_perm(v[3].y, v[3].x, 0x05040100u), __builtin_amdgcn_perm(v[3].w, v[3].z, 0x05040100u));
                    }
                    else {
                        uint4 *dstp = (uint4 *)(plane + (size_t)(i0 + cq) * 4);
#pragma unroll
                        for (int t = 0; t < 4; t++) dstp[t] = v[t];
                    }
                }
                else {
                    const uint32_t cc = f_ca[R];
                    int32_t *o = out + f_oo[R] * 2;
                    const bool p16 = tk < f_wf[R];
                    const uint32_t pw[8] = {pa[R][0].x, pa[R][0].y, pa[R][0].z, pa[R][0].w, pa[R][1].x, pa[R][1].y, pa[R][1].z, pa[R][1].w};
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const uint4 vb = lt[t], va = pa[R][t];
                        // (16-bit parking: value k of the lane's sixteen is half k & 1 of word k >> 1)
                        const uint32_t xa[4] = {p16 ? (uint32_t)((int32_t)(pw[2 * t] << 16) >> 16) : va.x, p16 ? (uint32_t)((int32_t)pw[2 * t] >> 16) : va.y,
                                                p16 ? (uint32_t)((int32_t)(pw[2 * t + 1] << 16) >> 16) : va.z, p16 ? (uint32_t)((int32_t)pw[2 * t + 1] >> 16) : va.w};
                        const uint32_t xb[4] = {vb.x, vb.y, vb.z, vb.w};
                        int32_t lo[4], ro[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int32_t av = (int32_t)xa[e], bv = (int32_t)(xb[e] << wsh);
                            int32_t ma, mb;
                            if (WIDE) {
                                const i64 side = (i64)((u64)(i64)(int32_t)xb[e] << wsh);
                                const i64 mid = (i64)(((u64)(i64)av) << 1) | (side & 1);
                                ma = (int32_t)((mid + side) >> 1); mb = (int32_t)((mid - side) >> 1);
                            }
                            else {
                                // (up to 16-bit samples: mid and side need 18 bits)
                                const int32_t mid = (int32_t)(((uint32_t)av << 1) | ((uint32_t)bv & 1));
                                ma = (mid + bv) >> 1; mb = (mid - bv) >> 1;
                            }
                            lo[e] = cc == 2 ? av + bv : cc == 3 ? ma : av;
                            ro[e] = cc == 1 ? av - bv : cc == 3 ? mb : bv;
                        }
                        const uint32_t i = i0 + cq + 4 * t;
                        if (interleave & 1) {
                            int4 *dd = (int4 *)(o + (size_t)i * 2);
                            dd[0] = make_int4(lo[0], ro[0], lo[1], ro[1]);
                            dd[1] = make_int4(lo[2], ro[2], lo[3], ro[3]);
                        }
                        else {
                            *(int4 *)(o + i) = make_int4(lo[0], lo[1], lo[2], lo[3]);
                            *(int4 *)(o + f_n[R] + i) = make_int4(ro[0], ro[1], ro[2], ro[3]);
                        }
                    }
                }
            }
            // request channel 0 of the next tile (the first tile of channel 1 after the last of channel 0, or the next one
            // of channel 1): the load then has a whole tile to arrive.  Every lane loads in both rounds -- from the start of
            // the plane when its row has nothing to fetch --, so that no load has to be merged with an old value (the
            // compiler would wait for it on the spot, one memory latency per round).
            {
                const uint32_t itn = it + 1, chn = itn / tpc, i0n = (itn % tpc) * FG_TS;
                if (itn < T && chn == 1) {
#pragma unroll
                    for (int R = 0; R < NR; R++) {
                        const uint32_t row = R * 16 + ((uint32_t)lane >> 2);
                        const bool want = (uint32_t)R < nrnd && row < G && f_c[R] == 2 && i0n + FG_TS <= f_n[R];
                        const bool p16 = (itn % tpc) < f_wf[R];
                        const uint4 *src = (const uint4 *)(want ? (const char *)(scratch + f_oo[R] * 2) + (size_t)(i0n + cq) * (p16 ? 2 : 4) : (const char *)scratch);
                        const uint4 *src2 = (want && !p16) ? src : (const uint4 *)scratch;       // (16-bit parking: the second half is not needed)
                        pa[R][0] = src[0]; pa[R][1] = src[1]; pa[R][2] = src2[2]; pa[R][3] = src2[3];
                        pa_it[R] = want ? itn : ~0u;
                    }
                }
            }
            continue;
        }
        // ---- general form (32-bit parking, and for the rest of the frame)
        if (ch == 0) {
#pragma unroll
            for (int R = 0; R < NR; R++) f_wf[R] = f_wf[R] < tk ? f_wf[R] : tk;
        }
        const uint32_t i = i0 + (uint32_t)lane;
        for (uint32_t r = 0; r < G; r++) {
            const uint32_t *fm = L.frm + r * FG_FRM;
            const uint32_t rn = rn_[r], C = fm[1];
            if (i0 >= rn || ch >= C) continue;
            const uint32_t nfr = fm[0], ca = fm[2];
            const u64 oo = ((u64)fm[4] << 32) | fm[3];
            const uint32_t wasted = L.subp[((ch & 3) * G + r) * FG_SUBP + 2];
            const uint32_t x = tile[r * FG_TSTR + lane];
            if (i >= rn) continue;
            if (C == 2) {
                if (ch == 0) scratch[oo * 2 + i] = (int32_t)(x << wasted);           // parked until the second channel arrives
                else {
                    // (the tile's format is the row's, kept by the lanes of the fast form: 16 bits below f_wf)
                    uint32_t wf = 0;
#pragma unroll
                    for (int R = 0; R < NR; R++) if ((r >> 4) == (uint32_t)R) wf = (uint32_t)__builtin_amdgcn_readlane((int)f_wf[R], (int)((r & 15) * 4));
                    const int32_t av = tk < wf ? (int32_t)((const int16_t *)(scratch + oo * 2))[i] : scratch[oo * 2 + i], bv = (int32_t)(x << wasted);
                    // (32-bit streams: a side channel with wasted bits is a 33-bit value once shifted back)
                    const i64 side = WIDE ? (i64)((u64)(i64)(int32_t)x << wasted) : (i64)bv;
                    const i64 mid = (i64)(((u64)(i64)av) << 1) | (side & 1);
                    const int32_t ma = (int32_t)((mid + side) >> 1), mb = (int32_t)((mid - side) >> 1);
                    const int32_t lo = ca == 2 ? av + bv : ca == 3 ? ma : av;
                    const int32_t ro = ca == 1 ? av - bv : ca == 3 ? mb : bv;
                    int32_t *o = out + oo * 2;
                    if (interleave & 1) ((int2 *)o)[i] = make_int2(lo, ro);
                    else { o[i] = lo; o[nfr + i] = ro; }
                }
            }
            else {
                int32_t *o = out + oo * C;
                const int32_t v = (int32_t)(x << wasted);
                if (interleave & 1) o[(size_t)i * C + ch] = v;
                else o[(size_t)ch * nfr + i] = v;
            }
        }
    }
    FG_PROF_END();
    if (prof && lane == 0) { u64 *pp_ = prof + ((size_t)((nframes + 63) & ~63u) + blockIdx.x) * 8; pp_[0] = pw_busy[0]; pp_[1] = pw_busy[1]; }
#undef FG_BAR
#undef FG_PROF_END
}

#endif  // FG_LEGACY
// Settle the frames after the fused kernel and the CRC-16 kernel: merge the CRC verdict into the status, and write silence
// for frames that failed (libFLAC delivers silence on a CRC mismatch; status 3 = the generic kernel decodes it next).
__global__ void __launch_bounds__(256)
fg_dec_fix_kernel(const FgDecFrame *frames, uint32_t nframes, FgDecResult *results, int32_t *out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nframes) return;
    const FgDecFrame fr = frames[f];
    uint32_t status = 1, crcw = 0;
    if (fr.bytes != 0) {
        status = results[f].err;
        crcw = results[f].crc;
        if (status == 0 && (crcw & 0x80000000u)) status = 2;
    }
    if (fr.n != 0 && lane == 0) { results[f].err = status; results[f].crc = crcw & 0xFFFFu; }
    if (status != 0 && status != 3 && fr.n != 0 && fr.channels != 0) {
        int32_t *o = out + fr.out_off * fr.channels;
        for (uint32_t k = lane; k < fr.n * fr.channels; k += 64) o[k] = 0;
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

// Frames per wave.  A lane's work is one serial chain, so a wave takes as long as its slowest frame however many lanes are
// busy; a launch that cannot fill the chip is spread thin.  Measured on the MI355X with 7032 frames (tools/gpu_gsweep.sh):
// the parse kernel is fastest with about two waves per CU (14-20 frames per wave: 0.57 ms against 0.63 ms at four waves
// per CU and 0.82 ms at eight), the restore kernel with four per CU.  Large batches fill the lanes (G up to the LDS limit).
static uint32_t fg_dec_group(uint32_t nframes, uint32_t per_frame_lanes, uint32_t waves_per_cu)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    uint32_t slots = (uint32_t)cus * waves_per_cu;
    if (getenv("FLACGPU_DEC_WPS")) slots = (uint32_t)cus * 4 * (uint32_t)atoi(getenv("FLACGPU_DEC_WPS"));     // tuning aid: waves per SIMD
    uint32_t g = (nframes + slots - 1) / slots;
    const uint32_t gmax = 64 / per_frame_lanes;
    if (g < 1) g = 1;
    if (g > gmax) g = gmax;
    return g;
}

#ifndef FG_LEGACY
// The lane-serial decoders of rounds 1 and 2 (fg_dec_rice_kernel + fg_dec_restore_kernel, fg_dec_fused_kernel) are built with
// `make LEGACY=1` only; the wave-parallel parser and its restore kernel (flac_dec_wave.hip) replace them.
extern "C" int fg_launch_decode_fast(const uint8_t *, uint64_t, const FgDecFrame *, uint32_t, int32_t *, FgDecSub *, FgDecResult *, int,
                                     unsigned long long *, uint16_t *, hipStream_t) { return -3; }
extern "C" int fg_launch_decode_finish(const uint8_t *, const FgDecFrame *, uint32_t, uint32_t, const int32_t *, const FgDecSub *, int32_t *,
                                       FgDecResult *, const uint16_t *, uint32_t, int, unsigned long long *, hipStream_t) { return -3; }
extern "C" int fg_launch_decode_fused(const uint8_t *, uint64_t, const FgDecFrame *, uint32_t, int32_t *, FgDecSub *, FgDecResult *, int,
                                      uint16_t *, int32_t *, int32_t *, uint32_t, unsigned long long *, hipStream_t) { return -3; }
#else
extern "C" int fg_launch_decode_fast(const uint8_t *d_stream, uint64_t stream_len, const FgDecFrame *d_frames, uint32_t nframes,
                                     int32_t *d_scratch, FgDecSub *d_subs, FgDecResult *d_results, int wide, unsigned long long *d_prof,
                                     uint16_t *d_rparams, hipStream_t stream)
{
    if (nframes == 0) return 0;
    uint32_t G = fg_dec_group(nframes, 1, 2);
    if (getenv("FLACGPU_DEC_G1")) G = (uint32_t)atoi(getenv("FLACGPU_DEC_G1"));      // tuning aid
    if (G > 32) G = 32;     // LDS per wave grows with G (ring + tile rows); 32 keeps several waves per CU
    const size_t lds = ((size_t)G * (FG_RSTR + 2 * FG_TSTR) + 2 * 64 * FG_META + 4) * 4;
    hipLaunchKernelGGL(fg_dec_rice_kernel, dim3((nframes + G - 1) / G), dim3(128), lds, stream, d_stream, (u64)stream_len, d_frames, nframes, G,
                       wide ? 0u : 1u, d_scratch, d_subs, d_results, d_prof, d_rparams);
    return (int)hipGetLastError();
}

extern "C" int fg_launch_decode_finish(const uint8_t *d_stream, const FgDecFrame *d_frames, uint32_t nframes, uint32_t channels,
                                       const int32_t *d_scratch, const FgDecSub *d_subs, int32_t *d_pcm, FgDecResult *d_results,
                                       const uint16_t *d_crctab, uint32_t interleave, int wide, unsigned long long *d_prof, hipStream_t stream)
{
    if (nframes == 0) return 0;
    const uint32_t C = channels ? channels : 1;
    uint32_t G = fg_dec_group(nframes, C, 4);
    if (getenv("FLACGPU_DEC_G2")) G = (uint32_t)atoi(getenv("FLACGPU_DEC_G2"));      // tuning aid
    if (G * C > FG_RROWS) G = FG_RROWS / C;
    if (G < 1) return -1;
    const dim3 grid((nframes + G - 1) / G);
    const size_t lds = (size_t)(((G * C + 15) & ~15u) + 1) * FG_TRS * 4;      // rounds of 16 rows + the spare row
    if (wide) hipLaunchKernelGGL(fg_dec_restore_kernel<true>, grid, dim3(128), lds, stream, d_frames, nframes, G, C, d_subs, d_scratch, d_pcm, d_results, interleave, d_prof);
    else hipLaunchKernelGGL(fg_dec_restore_kernel<false>, grid, dim3(128), lds, stream, d_frames, nframes, G, C, d_subs, d_scratch, d_pcm, d_results, interleave, d_prof);
    return (int)hipGetLastError();
}

// The fused decoder (parse + convert + recurrence + output in one kernel); fg_launch_decode_fix after it and the CRC kernel.
extern "C" int fg_launch_decode_fused(const uint8_t *d_stream, uint64_t stream_len, const FgDecFrame *d_frames, uint32_t nframes,
                                      int32_t *d_scratch, FgDecSub *d_subs, FgDecResult *d_results, int wide, uint16_t *d_rparams,
                                      int32_t *d_warm, int32_t *d_pcm, uint32_t interleave, unsigned long long *d_prof, hipStream_t stream)
{
    if (nframes == 0) return 0;
    // one workgroup (four waves, one per SIMD) per CU while the frames allow: 28 frames per group for the 7032 frames of a
    // 600 s stream (0.51 ms; 0.56 ms with two groups of 14 per CU, which share the SIMDs)
    uint32_t G = fg_dec_group(nframes, 1, 1);
    // Large batches: the workgroup's LDS (ring, tiles and residual tiles per frame) holds 48 frames, and a CU holds one
    // workgroup, so the frames are spread evenly over the fewest passes of 48 per CU (90 112 frames: 8 passes of 44).
    if (G > 32) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const uint32_t per_pass = (uint32_t)cus * FG_FUSED_GMAX;
        const uint32_t passes = (nframes + per_pass - 1) / per_pass;
        G = (nframes + (uint32_t)cus * passes - 1) / ((uint32_t)cus * passes);
        if (G < 32) G = 32;
    }
    if (getenv("FLACGPU_DEC_G1")) G = (uint32_t)atoi(getenv("FLACGPU_DEC_G1"));      // tuning aid
    if (G > FG_FUSED_GMAX) G = FG_FUSED_GMAX;
    const size_t lds = ((size_t)G * (FG_RSTR + 2 * FG_TSTR) + 2 * 64 * FG_META + 8 + (size_t)FG_RT * (G + 1) * FG_TSTR + FG_RT * 64 +
                        4 * (size_t)G * FG_SUBP + (size_t)G * FG_FRM + 192) * 4;
    const int which = (wide ? 1 : 0) + (G > 32 ? 2 : 0);
    const void *fn = which == 0 ? (const void *)fg_dec_fused_kernel<false, 2> : which == 1 ? (const void *)fg_dec_fused_kernel<true, 2>
                   : which == 2 ? (const void *)fg_dec_fused_kernel<false, 3> : (const void *)fg_dec_fused_kernel<true, 3>;
    if (fg_func_set_lds(fn, lds) != 0) return -1;
    if (fg_tune("FLACGPU_DEC_SKIP")) interleave |= ((uint32_t)atoi(fg_tune("FLACGPU_DEC_SKIP")) & 3u) << 8;     // experiments: 1 = no output wave, 2 = no recurrence
    const dim3 grid((nframes + G - 1) / G);
#define FG_FUSED_LAUNCH(W, N) hipLaunchKernelGGL((fg_dec_fused_kernel<W, N>), grid, dim3(256), lds, stream, d_stream, (u64)stream_len, d_frames, nframes, G, \
                                                 W ? 0u : 1u, d_scratch, d_subs, d_results, d_rparams, d_warm, d_pcm, interleave, (u64 *)d_prof)
    switch (which) {
    case 0: FG_FUSED_LAUNCH(false, 2); break;
    case 1: FG_FUSED_LAUNCH(true, 2); break;
    case 2: FG_FUSED_LAUNCH(false, 3); break;
    default: FG_FUSED_LAUNCH(true, 3); break;
    }
#undef FG_FUSED_LAUNCH
    return (int)hipGetLastError();
}

#endif  // FG_LEGACY
extern "C" int fg_launch_decode_fix(const FgDecFrame *d_frames, uint32_t nframes, FgDecResult *d_results, int32_t *d_pcm, hipStream_t stream)
{
    if (nframes == 0) return 0;
    hipLaunchKernelGGL(fg_dec_fix_kernel, dim3((nframes + 3) / 4), dim3(256), 0, stream, d_frames, nframes, d_results, d_pcm);
    return (int)hipGetLastError();
}

// CRC-16 of every frame; independent of the parse kernel, so the caller may run it on a second stream beside it
extern "C" int fg_launch_decode_crc(const uint8_t *d_stream, const FgDecFrame *d_frames, uint32_t nframes, FgDecResult *d_results,
                                    const uint16_t *d_crctab, hipStream_t stream, const unsigned long long *d_offsets, unsigned long long stream_len)
{
    if (nframes == 0) return 0;
    hipLaunchKernelGGL(fg_dec_crc_kernel, dim3((nframes + 3) / 4), dim3(256), 0, stream, d_stream, d_frames, nframes, d_results, d_crctab,
                       (const u64 *)d_offsets, (u64)stream_len);
    return (int)hipGetLastError();
}
