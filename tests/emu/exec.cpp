// gfx950emu (test infrastructure, see emu.h): one instruction of one wave.
#include "exec.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>

void emu_fault(Wave &w, const char *fmt, ...)
{
    char msg[512];
    va_list ap; va_start(ap, fmt); vsnprintf(msg, sizeof msg, fmt, ap); va_end(ap);
    const Inst &in = w.d->co->insts[w.pc];
    char full[1024];
    snprintf(full, sizeof full, "%s: %s at %s+0x%llx (disassembly line %u, %s), workgroup %u", w.d->ki->name.c_str(), msg, "code",
             (unsigned long long)in.addr, in.line, op_name(in.op), w.wg->id[0]);
    if (!w.d->failed) { w.d->failed = true; w.d->error = full; fprintf(stderr, "gfx950emu: FAULT %s\n", full); }
    w.state = W_FAULT;
}

static inline u32 f2u(float f) { u32 u; memcpy(&u, &f, 4); return u; }
static inline float u2f(u32 u) { float f; memcpy(&f, &u, 4); return f; }
static inline u64 d2u(double d) { u64 u; memcpy(&u, &d, 8); return u; }
static inline double u2d(u64 u) { double d; memcpy(&d, &u, 8); return d; }
static inline i32 sext24(u32 v) { return (i32)(v << 8) >> 8; }
static inline u32 brev32(u32 v) { u32 r = 0; for (int k = 0; k < 32; k++) if (v >> k & 1) r |= 1u << (31 - k); return r; }
static inline u64 brev64(u64 v) { return ((u64)brev32((u32)v) << 32) | brev32((u32)(v >> 32)); }

// ---------------------------------------------------------------------------------------------------------------- scalar operands
static inline u32 rs32(Wave &w, const Opnd &o)
{
    switch (o.kind) {
    case K_SGPR: return w.s[o.reg];
    case K_VCC: case K_VCC_LO: return (u32)w.vcc;
    case K_VCC_HI: return (u32)(w.vcc >> 32);
    case K_EXEC: case K_EXEC_LO: return (u32)w.exec;
    case K_EXEC_HI: return (u32)(w.exec >> 32);
    case K_M0: return w.m0;
    case K_SCC: return w.scc ? 1u : 0u;
    case K_IMM: return (u32)o.imm;
    case K_FIMM: return f2u((float)o.f);
    default: return 0;
    }
}
static inline u64 rs64(Wave &w, const Opnd &o)
{
    switch (o.kind) {
    case K_SGPR: return (u64)w.s[o.reg] | ((u64)w.s[o.reg + 1] << 32);
    case K_VCC: return w.vcc;
    case K_EXEC: return w.exec;
    case K_IMM: return (u64)o.imm;           // (inline constants sign-extend; a 32-bit literal of a b64 operand is what the printer shows)
    case K_FIMM: return d2u(o.f);
    case K_SHARED_BASE: return (u64)EMU_SHARED_HI << 32;
    case K_PRIVATE_BASE: return (u64)EMU_PRIVATE_HI << 32;
    case K_SHARED_LIMIT: return ((u64)EMU_SHARED_HI << 32) | 0xFFFFFFFFull;
    case K_PRIVATE_LIMIT: return ((u64)EMU_PRIVATE_HI << 32) | 0xFFFFFFFFull;
    default: return rs32(w, o);
    }
}
static inline void ws32(Wave &w, const Opnd &o, u32 v)
{
    switch (o.kind) {
    case K_SGPR: w.s[o.reg] = v; break;
    case K_VCC: w.vcc = v; break;
    case K_VCC_LO: w.vcc = (w.vcc & ~0xFFFFFFFFull) | v; break;
    case K_VCC_HI: w.vcc = (w.vcc & 0xFFFFFFFFull) | ((u64)v << 32); break;
    case K_EXEC: w.exec = v; break;
    case K_EXEC_LO: w.exec = (w.exec & ~0xFFFFFFFFull) | v; break;
    case K_EXEC_HI: w.exec = (w.exec & 0xFFFFFFFFull) | ((u64)v << 32); break;
    case K_M0: w.m0 = v; break;
    default: break;
    }
}
static inline void ws64(Wave &w, const Opnd &o, u64 v)
{
    switch (o.kind) {
    case K_SGPR: w.s[o.reg] = (u32)v; w.s[o.reg + 1] = (u32)(v >> 32); break;
    case K_VCC: w.vcc = v; break;
    case K_EXEC: w.exec = v; break;
    default: ws32(w, o, (u32)v); break;
    }
}

// ---------------------------------------------------------------------------------------------------------------- memory
static inline bool glob_ok(Wave &w, u64 a, u32 n)
{
    if (emu_mem_ok(a, n)) return true;
    emu_fault(w, "access of %u bytes at 0x%llx outside every allocation", n, (unsigned long long)a);
    return false;
}
// a flat address: LDS, scratch or global
static inline u8 *flat_ptr(Wave &w, u64 a, u32 n, int lane)
{
    const u32 hi = (u32)(a >> 32), lo = (u32)a;
    if (hi == EMU_SHARED_HI) {
        if ((u64)lo + n > w.wg->lds.size()) { emu_fault(w, "flat access of LDS at %u beyond %zu", lo, w.wg->lds.size()); return nullptr; }
        return w.wg->lds.data() + lo;
    }
    if (hi == EMU_PRIVATE_HI) {
        const u32 sz = w.d->ki->scratch;
        if ((u64)lo + n > sz) { emu_fault(w, "flat access of scratch at %u beyond %u", lo, sz); return nullptr; }
        return w.scratch.data() + (size_t)lane * sz + lo;
    }
    if (!glob_ok(w, a, n)) return nullptr;
    return (u8 *)(uintptr_t)a;
}

static inline u32 dpp_src_lane(u16 c, int l, bool *valid)
{
    *valid = true;
    const int row = l & ~15, r = l & 15;
    if (c <= 0xFF) return (u32)((l & ~3) | ((c >> (2 * (l & 3))) & 3));
    if (c >= 0x101 && c <= 0x10F) { const int n = c & 15; if (r + n > 15) *valid = false; return (u32)(l + n) & 63; }
    if (c >= 0x111 && c <= 0x11F) { const int n = c & 15; if (r < n) *valid = false; return (u32)(l - n) & 63; }
    if (c >= 0x121 && c <= 0x12F) { const int n = c & 15; return (u32)(row | ((r - n) & 15)); }
    switch (c) {
    case 0x130: if (l == 63) *valid = false; return (u32)(l + 1) & 63;
    case 0x134: return (u32)(l + 1) & 63;
    case 0x138: if (l == 0) *valid = false; return (u32)(l - 1) & 63;
    case 0x13C: return (u32)(l - 1) & 63;
    case 0x140: return (u32)(row | (15 - r));
    case 0x141: return (u32)((l & ~7) | (7 - (l & 7)));
    case 0x142: if (l < 16) { *valid = false; return 0; } return (u32)(((l >> 4) - 1) * 16 + 15);
    case 0x143: if (l < 32) { *valid = false; return 0; } return 31;
    }
    if (c >= 0x150 && c <= 0x15F) return (u32)(row | (c & 15));
    *valid = false;
    return 0;
}

namespace {
struct VX {
    Wave &w;
    const Inst &in;
    int sb;                 // index of the first source operand
    u64 wm;                 // lanes that write their result
    u32 dppv[64];
    u64 dppv64[64];
    bool dpp;
    VX(Wave &w_, const Inst &in_, int sb_) : w(w_), in(in_), sb(sb_), wm(w_.exec), dpp(false) {}

    inline u32 vreg(const Opnd &o, int k, int l, int dw = 0) const
    {
        u32 r = o.reg + dw;
        if (w.gpr_idx_mode && (w.gpr_idx_mode >> k & 1)) r += w.gpr_idx;
        return w.v[(size_t)r * 64 + l];
    }
    inline u32 raw32(int k, int l) const
    {
        const Opnd &o = in.o[sb + k];
        if (o.kind == K_VGPR) return vreg(o, k, l);
        if (o.kind == K_AGPR) return w.a[(size_t)o.reg * 64 + l];
        return rs32(w, o);
    }
    inline u64 raw64(int k, int l, bool f64) const
    {
        const Opnd &o = in.o[sb + k];
        if (o.kind == K_VGPR) return (u64)vreg(o, k, l) | ((u64)vreg(o, k, l, 1) << 32);
        if (o.kind == K_AGPR) return (u64)w.a[(size_t)o.reg * 64 + l] | ((u64)w.a[(size_t)(o.reg + 1) * 64 + l] << 32);
        if (o.kind == K_IMM) {
            // inline constants -16 .. 64 are the 64-bit integer; a 32-bit literal is the HIGH half of an f64 operand, the zero-extended
            // low half of an integer one
            if (o.imm >= -16 && o.imm <= 64) return (u64)o.imm;
            return f64 ? ((u64)(u32)o.imm << 32) : (u64)(u32)o.imm;
        }
        return rs64(w, o);
    }
    static inline u32 sel32(u32 v, u8 sel, bool sx)
    {
        switch (sel) {
        case SEL_BYTE0: v = v & 0xFF; return sx ? (u32)(i32)(i8)v : v;
        case SEL_BYTE1: v = (v >> 8) & 0xFF; return sx ? (u32)(i32)(i8)v : v;
        case SEL_BYTE2: v = (v >> 16) & 0xFF; return sx ? (u32)(i32)(i8)v : v;
        case SEL_BYTE3: v = v >> 24; return sx ? (u32)(i32)(i8)v : v;
        case SEL_WORD0: v = v & 0xFFFF; return sx ? (u32)(i32)(i16)v : v;
        case SEL_WORD1: v = v >> 16; return sx ? (u32)(i32)(i16)v : v;
        default: return v;
        }
    }
    // integer source
    inline u32 S(int k, int l) const
    {
        u32 v = (k == 0 && dpp) ? dppv[l] : raw32(k, l);
        if (in.enc == E_SDWA && k < 2) v = sel32(v, k == 0 ? in.src0_sel : in.src1_sel, (in.o[sb + k].flags & F_SEXT) != 0);
        return v;
    }
    // f32 source (|x|, -x)
    inline float F(int k, int l) const
    {
        u32 v = S(k, l);
        const u8 fl = in.o[sb + k].flags;
        if (fl & F_ABS) v &= 0x7FFFFFFFu;
        if (fl & F_NEG) v ^= 0x80000000u;
        return u2f(v);
    }
    inline u64 S64(int k, int l) const { return (k == 0 && dpp) ? dppv64[l] : raw64(k, l, false); }
    inline double D(int k, int l) const
    {
        u64 v = (k == 0 && dpp) ? dppv64[l] : raw64(k, l, true);
        const u8 fl = in.o[sb + k].flags;
        if (fl & F_ABS) v &= 0x7FFFFFFFFFFFFFFFull;
        if (fl & F_NEG) v ^= 0x8000000000000000ull;
        return u2d(v);
    }
    inline u32 dreg(int dw = 0) const
    {
        u32 r = in.o[0].reg + dw;
        if (w.gpr_idx_mode & 8) r += w.gpr_idx;
        return r;
    }
    inline void W(int l, u32 v) const
    {
        if (in.o[0].kind == K_AGPR) { w.a[(size_t)in.o[0].reg * 64 + l] = v; return; }
        u32 &d = w.v[(size_t)dreg() * 64 + l];
        if (in.enc == E_SDWA && in.dst_sel != SEL_DWORD) {
            int sh, bits;
            switch (in.dst_sel) { case SEL_BYTE0: sh = 0; bits = 8; break; case SEL_BYTE1: sh = 8; bits = 8; break; case SEL_BYTE2: sh = 16; bits = 8; break;
                                  case SEL_BYTE3: sh = 24; bits = 8; break; case SEL_WORD0: sh = 0; bits = 16; break; default: sh = 16; bits = 16; break; }
            const u32 mask = ((1u << bits) - 1u) << sh, part = (v << sh) & mask;
            if (in.dst_unused == UNUSED_PRESERVE) d = (d & ~mask) | part;
            else if (in.dst_unused == UNUSED_SEXT) {
                const u32 sx = (part >> (sh + bits - 1)) & 1u ? ~0u : 0u;
                const u32 upper = (sh + bits >= 32) ? 0u : (sx << (sh + bits));
                d = part | upper;              // (bits below the field are zero)
            }
            else d = part;
            return;
        }
        d = v;
    }
    inline void W64(int l, u64 v) const
    {
        if (in.o[0].kind == K_AGPR) { w.a[(size_t)in.o[0].reg * 64 + l] = (u32)v; w.a[(size_t)(in.o[0].reg + 1) * 64 + l] = (u32)(v >> 32); return; }
        w.v[(size_t)dreg() * 64 + l] = (u32)v; w.v[(size_t)dreg(1) * 64 + l] = (u32)(v >> 32);
    }
    inline void WF(int l, float f) const { W(l, f2u(f)); }
    inline void WD(int l, double d) const { W64(l, d2u(d)); }
    void setup_dpp(bool wide)
    {
        if (in.enc != E_DPP || in.dpp == 0xFFFF) return;
        dpp = true;
        u64 ok = 0;
        for (int l = 0; l < 64; l++) {
            bool valid;
            const u32 sl = dpp_src_lane(in.dpp, l, &valid);
            if (valid && !(w.exec >> sl & 1)) valid = false;          // a lane that is switched off has nothing to give
            const bool enabled = (in.row_mask >> (l >> 4) & 1) && (in.bank_mask >> ((l >> 2) & 3) & 1);
            u64 val = 0;
            if (valid) val = wide ? raw64(0, (int)sl, false) : (u64)raw32(0, (int)sl);
            else if (!in.bound_ctrl) { dppv[l] = 0; dppv64[l] = 0; continue; }        // the lane keeps what it has
            if (!enabled) continue;
            dppv[l] = (u32)val; dppv64[l] = val;
            ok |= 1ull << l;
        }
        wm &= ok;
    }
};
}  // namespace

#define VLOOP for (int l = 0; l < 64; l++) if (x.wm >> l & 1)

static inline int cls_f64(double v)
{
    const u64 u = d2u(v); const bool neg = u >> 63;
    const u64 ex = (u >> 52) & 0x7FF, man = u & 0xFFFFFFFFFFFFFull;
    if (ex == 0x7FF) { if (man) return (man >> 51) ? 1 : 0; return neg ? 2 : 9; }
    if (ex == 0) { if (man == 0) return neg ? 5 : 6; return neg ? 4 : 7; }
    return neg ? 3 : 8;
}
static inline int cls_f32(float v)
{
    const u32 u = f2u(v); const bool neg = u >> 31;
    const u32 ex = (u >> 23) & 0xFF, man = u & 0x7FFFFF;
    if (ex == 0xFF) { if (man) return (man >> 22) ? 1 : 0; return neg ? 2 : 9; }
    if (ex == 0) { if (man == 0) return neg ? 5 : 6; return neg ? 4 : 7; }
    return neg ? 3 : 8;
}
static inline int exp_f64(double v) { return (int)((d2u(v) >> 52) & 0x7FF); }

template <typename T> static inline bool cmp_op(int which, T a, T b)      // 0 lt 1 eq 2 le 3 gt 4 ne 5 ge
{
    switch (which) { case 0: return a < b; case 1: return a == b; case 2: return a <= b; case 3: return a > b; case 4: return a != b; default: return a >= b; }
}
// the sixteen float compares: F LT EQ LE GT LG GE O U NGE NLG NGT NLE NEQ NLT TRU
template <typename T> static inline bool fcmp_op(int which, T a, T b)
{
    const bool un = (a != a) || (b != b);
    switch (which) {
    case 0: return false; case 1: return a < b; case 2: return a == b; case 3: return a <= b; case 4: return a > b; case 5: return !un && a != b;
    case 6: return a >= b; case 7: return !un; case 8: return un; case 9: return !(a >= b); case 10: return un || a == b; case 11: return !(a > b);
    case 12: return !(a <= b); case 13: return !(a == b); case 14: return !(a < b); default: return true;
    }
}

static void exec_mfma_f64_4x4x4(Wave &w, const Inst &in)
{
    // D[b][i][j] = C[b][i][j] + sum_k A[b][i][k] B[b][k][j], the sum as a chain of fused multiply-adds in ascending k (what the
    // library's start-up self-check requires of the device).  A, B: lane k * 16 + b * 4 + x; C, D: lane i * 16 + b * 4 + j.
    VX x(w, in, 1);
    double A[64], B[64], C[64], D[64];
    for (int l = 0; l < 64; l++) { A[l] = u2d(x.raw64(0, l, true)); B[l] = u2d(x.raw64(1, l, true)); C[l] = u2d(x.raw64(2, l, true)); }
    for (int b = 0; b < 4; b++) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        double acc = C[i * 16 + b * 4 + j];
        for (int k = 0; k < 4; k++) acc = fma(A[k * 16 + b * 4 + i], B[k * 16 + b * 4 + j], acc);
        D[i * 16 + b * 4 + j] = acc;
    }
    for (int l = 0; l < 64; l++) x.W64(l, d2u(D[l]));
}

static double div_fixup_f64(double q, double den, double num)
{
    const bool sign = (d2u(den) ^ d2u(num)) >> 63;
    if (num != num) return num; if (den != den) return den;
    if (den == 0 && num == 0) return NAN;
    if (isinf(den) && isinf(num)) return NAN;
    if (den == 0 || isinf(num)) return sign ? -INFINITY : INFINITY;
    if (isinf(den) || num == 0) return sign ? -0.0 : 0.0;
    const int e = exp_f64(num) - exp_f64(den);
    if (e < -1075) return sign ? -0.0 : 0.0;
    if (e > 1024) return sign ? -INFINITY : INFINITY;
    return sign ? -fabs(q) : fabs(q);
}

static bool exec_valu(Wave &w, const Inst &in)
{
    const u16 op = in.op;
    auto mask_write = [&](const Opnd &o, u64 m) { if (o.kind == K_VCC || o.kind == K_EXEC || (o.kind == K_SGPR && o.n == 2)) ws64(w, o, m); else ws32(w, o, (u32)m); };
    // ---- compares: the mask goes to o[0]; lanes that are off read as 0
    {
        int base = -1, kind = 0;     // kind: 0 u32 1 i32 2 u64 3 i64 4 u16 5 i16 6 f64 7 f32
        if (op >= OP_v_cmp_lt_u32 && op <= OP_v_cmp_ge_u32) { base = OP_v_cmp_lt_u32; kind = 0; }
        else if (op >= OP_v_cmp_lt_i32 && op <= OP_v_cmp_ge_i32) { base = OP_v_cmp_lt_i32; kind = 1; }
        else if (op >= OP_v_cmp_lt_u64 && op <= OP_v_cmp_ge_u64) { base = OP_v_cmp_lt_u64; kind = 2; }
        else if (op >= OP_v_cmp_lt_i64 && op <= OP_v_cmp_ge_i64) { base = OP_v_cmp_lt_i64; kind = 3; }
        else if (op >= OP_v_cmp_lt_u16 && op <= OP_v_cmp_ge_u16) { base = OP_v_cmp_lt_u16; kind = 4; }
        else if (op >= OP_v_cmp_lt_i16 && op <= OP_v_cmp_ge_i16) { base = OP_v_cmp_lt_i16; kind = 5; }
        else if (op >= OP_v_cmp_f_f64 && op <= OP_v_cmp_tru_f64) { base = OP_v_cmp_f_f64; kind = 6; }
        else if (op >= OP_v_cmp_f_f32 && op <= OP_v_cmp_tru_f32) { base = OP_v_cmp_f_f32; kind = 7; }
        if (base >= 0) {
            VX x(w, in, 1);
            x.setup_dpp(kind == 2 || kind == 3 || kind == 6);
            const int which = op - base;
            u64 m = 0;
            VLOOP {
                bool r;
                switch (kind) {
                case 0: r = cmp_op<u32>(which, x.S(0, l), x.S(1, l)); break;
                case 1: r = cmp_op<i32>(which, (i32)x.S(0, l), (i32)x.S(1, l)); break;
                case 2: r = cmp_op<u64>(which, x.S64(0, l), x.S64(1, l)); break;
                case 3: r = cmp_op<i64>(which, (i64)x.S64(0, l), (i64)x.S64(1, l)); break;
                case 4: r = cmp_op<u16>(which, (u16)x.S(0, l), (u16)x.S(1, l)); break;
                case 5: r = cmp_op<i16>(which, (i16)x.S(0, l), (i16)x.S(1, l)); break;
                case 6: r = fcmp_op<double>(which, x.D(0, l), x.D(1, l)); break;
                default: r = fcmp_op<float>(which, x.F(0, l), x.F(1, l)); break;
                }
                if (r) m |= 1ull << l;
            }
            mask_write(in.o[0], m);
            return true;
        }
    }
    if (op == OP_v_cmp_class_f64 || op == OP_v_cmp_class_f32) {
        VX x(w, in, 1); u64 m = 0;
        VLOOP { const int c = op == OP_v_cmp_class_f64 ? cls_f64(x.D(0, l)) : cls_f32(x.F(0, l)); if (x.S(1, l) >> c & 1) m |= 1ull << l; }
        mask_write(in.o[0], m);
        return true;
    }
    // ---- carry forms: D, carry-out, S0, S1 [, carry-in]
    if (op >= OP_v_add_co_u32 && op <= OP_v_subbrev_co_u32) {
        VX x(w, in, 2);
        x.setup_dpp(false);
        const bool has_in = (op == OP_v_addc_co_u32 || op == OP_v_subb_co_u32 || op == OP_v_subbrev_co_u32);
        const u64 cin = has_in ? rs64(w, in.o[4]) : 0;
        u64 cout = 0;
        VLOOP {
            const u64 a = x.S(0, l), b = x.S(1, l), c = (cin >> l) & 1;
            u64 r;
            switch (op) {
            case OP_v_add_co_u32: r = a + b; break;
            case OP_v_addc_co_u32: r = a + b + c; break;
            case OP_v_sub_co_u32: r = a - b; break;
            case OP_v_subb_co_u32: r = a - b - c; break;
            case OP_v_subrev_co_u32: r = b - a; break;
            default: r = b - a - c; break;
            }
            if (r >> 32 & 1) cout |= 1ull << l;
            x.W(l, (u32)r);
        }
        mask_write(in.o[1], cout);
        return true;
    }
    if (op == OP_v_mad_u64_u32 || op == OP_v_mad_i64_i32) {
        VX x(w, in, 2); u64 cout = 0;
        VLOOP {
            const u64 c = x.S64(2, l);
            if (op == OP_v_mad_u64_u32) { const unsigned __int128 r = (unsigned __int128)x.S(0, l) * x.S(1, l) + c; if (r >> 64) cout |= 1ull << l; x.W64(l, (u64)r); }
            else { const __int128 r = (__int128)(i32)x.S(0, l) * (i32)x.S(1, l) + (__int128)(i64)c; x.W64(l, (u64)r); }
        }
        mask_write(in.o[1], cout);
        return true;
    }
    if (op == OP_v_div_scale_f64) {
        VX x(w, in, 2); u64 vcc = 0;
        VLOOP {
            const double s0 = x.D(0, l), s1 = x.D(1, l), s2 = x.D(2, l);
            double d = s0;
            const int e1 = exp_f64(s1), e2 = exp_f64(s2);
            if (s2 == 0 || s1 == 0 || s1 != s1 || s2 != s2) d = NAN;
            else if (e2 - e1 >= 768) { vcc |= 1ull << l; if (d2u(s0) == d2u(s1)) d = ldexp(s0, 128); }
            else if (e1 == 0) d = ldexp(s0, 128);
            else if (e1 >= 0x7FD && (e2 - e1) <= -1023) { vcc |= 1ull << l; if (d2u(s0) == d2u(s1)) d = ldexp(s0, -128); }
            else if (e1 >= 0x7FD) d = ldexp(s0, -128);
            else if (e2 - e1 <= -1023) { vcc |= 1ull << l; if (d2u(s0) == d2u(s2)) d = ldexp(s0, 128); }
            else if (e2 <= 53) d = ldexp(s0, 128);
            x.WD(l, d);
        }
        mask_write(in.o[1], vcc);
        return true;
    }
    if (op == OP_v_readlane_b32) { VX x(w, in, 1); const u32 sl = rs32(w, in.o[2]) & 63; ws32(w, in.o[0], x.raw32(0, (int)sl)); return true; }
    if (op == OP_v_readfirstlane_b32) { VX x(w, in, 1); const int sl = w.exec ? __builtin_ctzll(w.exec) : 0; ws32(w, in.o[0], x.raw32(0, sl)); return true; }
    if (op == OP_v_writelane_b32) { const u32 sl = rs32(w, in.o[2]) & 63; w.v[(size_t)in.o[0].reg * 64 + sl] = rs32(w, in.o[1]); return true; }
    if (op == OP_v_mfma_f64_4x4x4_4b_f64) { exec_mfma_f64_4x4x4(w, in); return true; }
    if (op == OP_v_nop) return true;

    VX x(w, in, 1);
    bool wide_src0 = false;
    switch (op) { case OP_v_mov_b64: case OP_v_lshl_add_u64: case OP_v_add_f64: case OP_v_mul_f64: case OP_v_fma_f64: case OP_v_fmac_f64: case OP_v_max_f64: case OP_v_min_f64: wide_src0 = true; break; default: break; }
    x.setup_dpp(wide_src0);
    switch (op) {
#define I1(NAME, EXPR) case OP_##NAME: VLOOP { const u32 a = x.S(0, l); (void)a; x.W(l, (u32)(EXPR)); } break;
#define I2(NAME, EXPR) case OP_##NAME: VLOOP { const u32 a = x.S(0, l), b = x.S(1, l); (void)a; (void)b; x.W(l, (u32)(EXPR)); } break;
#define I3(NAME, EXPR) case OP_##NAME: VLOOP { const u32 a = x.S(0, l), b = x.S(1, l), c = x.S(2, l); (void)a; (void)b; (void)c; x.W(l, (u32)(EXPR)); } break;
    I1(v_mov_b32, a)
    I1(v_accvgpr_read_b32, a) I1(v_accvgpr_write_b32, a) I1(v_accvgpr_mov_b32, a)
    case OP_v_mov_b64: VLOOP x.W64(l, x.S64(0, l)); break;
    // (clamp on the unsigned add / subtract saturates: what `x > 1 ? x - 1 : 0` compiles to)
    I2(v_add_u32, (in.clamp && a + b < a) ? 0xFFFFFFFFu : a + b) I2(v_sub_u32, (in.clamp && b > a) ? 0u : a - b) I2(v_subrev_u32, (in.clamp && a > b) ? 0u : b - a)
    I2(v_mul_lo_u32, a * b) I2(v_mul_hi_u32, ((u64)a * b) >> 32) I2(v_mul_hi_i32, (u64)((i64)(i32)a * (i32)b) >> 32)
    I2(v_mul_i32_i24, (i64)sext24(a) * sext24(b)) I2(v_mul_u32_u24, (u64)(a & 0xFFFFFF) * (b & 0xFFFFFF))
    I2(v_mul_hi_i32_i24, (u64)((i64)sext24(a) * sext24(b)) >> 32) I2(v_mul_hi_u32_u24, ((u64)(a & 0xFFFFFF) * (b & 0xFFFFFF)) >> 32)
    I3(v_mad_i32_i24, (u32)((i64)sext24(a) * sext24(b)) + c) I3(v_mad_u32_u24, (u32)((u64)(a & 0xFFFFFF) * (b & 0xFFFFFF)) + c)
    I2(v_and_b32, a & b) I2(v_or_b32, a | b) I2(v_xor_b32, a ^ b) I2(v_xnor_b32, ~(a ^ b)) I1(v_not_b32, ~a)
    I2(v_lshlrev_b32, b << (a & 31)) I2(v_lshrrev_b32, b >> (a & 31)) I2(v_ashrrev_i32, (i32)b >> (a & 31))
    case OP_v_lshlrev_b64: VLOOP x.W64(l, x.S64(1, l) << (x.S(0, l) & 63)); break;
    case OP_v_lshrrev_b64: VLOOP x.W64(l, x.S64(1, l) >> (x.S(0, l) & 63)); break;
    case OP_v_ashrrev_i64: VLOOP x.W64(l, (u64)((i64)x.S64(1, l) >> (x.S(0, l) & 63))); break;
    I3(v_lshl_add_u32, (a << (b & 31)) + c)
    case OP_v_lshl_add_u64: VLOOP x.W64(l, (x.S64(0, l) << (x.S(1, l) & 7)) + x.S64(2, l)); break;
    I3(v_add_lshl_u32, (a + b) << (c & 31)) I3(v_lshl_or_b32, (a << (b & 31)) | c) I3(v_and_or_b32, (a & b) | c) I3(v_or3_b32, a | b | c)
    I3(v_add3_u32, a + b + c) I3(v_xad_u32, (a ^ b) + c)
    I3(v_bfe_u32, (c & 31) ? ((a >> (b & 31)) & ((1u << (c & 31)) - 1u)) : 0u)
    case OP_v_bfe_i32: VLOOP { const u32 a = x.S(0, l), off = x.S(1, l) & 31, wd = x.S(2, l) & 31;
                                u32 r = 0; if (wd) { const u32 f = (a >> off) & ((1u << wd) - 1u); r = (u32)((i32)(f << (32 - wd)) >> (32 - wd)); } x.W(l, r); } break;
    I3(v_bfi_b32, (a & b) | (~a & c)) I2(v_bfm_b32, ((1u << (a & 31)) - 1u) << (b & 31))
    I1(v_bfrev_b32, brev32(a))
    I3(v_alignbit_b32, (((u64)a << 32) | b) >> (c & 31)) I3(v_alignbyte_b32, (((u64)a << 32) | b) >> (8 * (c & 3)))
    case OP_v_perm_b32: VLOOP { const u32 a = x.S(0, l), b = x.S(1, l), c = x.S(2, l); const u64 src = ((u64)a << 32) | b; u32 r = 0;
                                 for (int k = 0; k < 4; k++) { const u32 sel = (c >> (8 * k)) & 0xFF; u32 byte;
                                     if (sel <= 7) byte = (u32)(src >> (8 * sel)) & 0xFF;
                                     else if (sel == 8) byte = (b >> 15 & 1) ? 0xFF : 0; else if (sel == 9) byte = (b >> 31 & 1) ? 0xFF : 0;
                                     else if (sel == 10) byte = (a >> 15 & 1) ? 0xFF : 0; else if (sel == 11) byte = (a >> 31 & 1) ? 0xFF : 0;
                                     else if (sel == 12) byte = 0; else byte = 0xFF;
                                     r |= byte << (8 * k); }
                                 x.W(l, r); } break;
    case OP_v_cndmask_b32: { const u64 m = in.no >= 4 ? rs64(w, in.o[3]) : w.vcc; VLOOP x.W(l, (m >> l & 1) ? x.S(1, l) : x.S(0, l)); } break;
    I2(v_min_u32, a < b ? a : b) I2(v_max_u32, a > b ? a : b) I2(v_min_i32, (i32)a < (i32)b ? a : b) I2(v_max_i32, (i32)a > (i32)b ? a : b)
    I3(v_min3_u32, std::min(a, std::min(b, c))) I3(v_max3_u32, std::max(a, std::max(b, c)))
    I3(v_med3_u32, std::max(std::min(a, b), std::min(std::max(a, b), c)))
    I3(v_min3_i32, std::min((i32)a, std::min((i32)b, (i32)c))) I3(v_max3_i32, std::max((i32)a, std::max((i32)b, (i32)c)))
    I3(v_med3_i32, std::max(std::min((i32)a, (i32)b), std::min(std::max((i32)a, (i32)b), (i32)c)))
    I3(v_sad_u32, (a > b ? a - b : b - a) + c)
    I1(v_ffbh_u32, a ? (u32)__builtin_clz(a) : 0xFFFFFFFFu) I1(v_ffbl_b32, a ? (u32)__builtin_ctz(a) : 0xFFFFFFFFu)
    I1(v_ffbh_i32, (a == 0 || a == 0xFFFFFFFFu) ? 0xFFFFFFFFu : (u32)__builtin_clz((i32)a < 0 ? ~a : a))
    I2(v_bcnt_u32_b32, (u32)__builtin_popcount(a) + b)
    case OP_v_mbcnt_lo_u32_b32: VLOOP { const u32 m = l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u); x.W(l, (u32)__builtin_popcount(x.S(0, l) & m) + x.S(1, l)); } break;
    case OP_v_mbcnt_hi_u32_b32: VLOOP { const u32 m = l <= 32 ? 0u : ((1u << (l - 32)) - 1u); x.W(l, (u32)__builtin_popcount(x.S(0, l) & m) + x.S(1, l)); } break;
    case OP_v_bitop3_b32: case OP_v_bitop3_b16: VLOOP { const u32 a = x.S(0, l), b = x.S(1, l), c = x.S(2, l); u32 r = 0;
                                for (int k = 0; k < 8; k++) if (in.bitop3 >> k & 1) r |= ((k & 4) ? a : ~a) & ((k & 2) ? b : ~b) & ((k & 1) ? c : ~c);
                                x.W(l, op == OP_v_bitop3_b16 ? (r & 0xFFFF) : r); } break;
    I3(v_dot2_i32_i16, (u32)((i32)(i16)a * (i32)(i16)b + (i32)(i16)(a >> 16) * (i32)(i16)(b >> 16)) + c)
    // (v_pk_mov_b32: D.lo = op_sel[0] ? S0.hi : S0.lo, D.hi = op_sel[1] ? S1.hi : S1.lo)
    case OP_v_pk_mov_b32: VLOOP { const u64 a = x.S64(0, l), b = x.S64(1, l);
                                   const u32 lo = (in.op_sel & 1) ? (u32)(a >> 32) : (u32)a, hi = (in.op_sel & 2) ? (u32)(b >> 32) : (u32)b; x.W64(l, (u64)lo | ((u64)hi << 32)); } break;
    I2(v_lshlrev_b16, (u16)((u16)b << (a & 15))) I2(v_lshrrev_b16, (u16)((u16)b >> (a & 15))) I2(v_ashrrev_i16, (u16)((i16)b >> (a & 15)))
    I2(v_add_u16, (u16)(a + b)) I2(v_sub_u16, (u16)(a - b)) I2(v_mul_lo_u16, (u16)(a * b)) I3(v_mad_u16, (u16)(a * b + c))
    I2(v_max_u16, std::max((u16)a, (u16)b)) I2(v_min_u16, std::min((u16)a, (u16)b)) I2(v_max_i16, (u16)std::max((i16)a, (i16)b)) I2(v_min_i16, (u16)std::min((i16)a, (i16)b))
    // ---- f32
    case OP_v_cvt_f32_u32: VLOOP x.WF(l, (float)x.S(0, l)); break;
    case OP_v_cvt_f32_i32: VLOOP x.WF(l, (float)(i32)x.S(0, l)); break;
    case OP_v_cvt_u32_f32: VLOOP { const float f = x.F(0, l); x.W(l, f != f ? 0u : f <= 0.f ? 0u : f >= 4294967296.f ? 0xFFFFFFFFu : (u32)f); } break;
    case OP_v_cvt_i32_f32: VLOOP { const float f = x.F(0, l); x.W(l, f != f ? 0u : f <= -2147483648.f ? 0x80000000u : f >= 2147483648.f ? 0x7FFFFFFFu : (u32)(i32)f); } break;
    case OP_v_rcp_iflag_f32: case OP_v_rcp_f32: VLOOP x.WF(l, 1.0f / x.F(0, l)); break;
    case OP_v_ldexp_f32: VLOOP x.WF(l, ldexpf(x.F(0, l), (i32)x.S(1, l))); break;
    case OP_v_cvt_f32_ubyte0: VLOOP x.WF(l, (float)(x.S(0, l) & 0xFF)); break;
    case OP_v_cvt_f32_ubyte1: VLOOP x.WF(l, (float)((x.S(0, l) >> 8) & 0xFF)); break;
    case OP_v_cvt_f32_ubyte2: VLOOP x.WF(l, (float)((x.S(0, l) >> 16) & 0xFF)); break;
    case OP_v_cvt_f32_ubyte3: VLOOP x.WF(l, (float)(x.S(0, l) >> 24)); break;
    case OP_v_cvt_f32_f64: VLOOP x.WF(l, (float)x.D(0, l)); break;
    case OP_v_cvt_f64_f32: VLOOP x.WD(l, (double)x.F(0, l)); break;
    case OP_v_mul_f32: VLOOP x.WF(l, x.F(0, l) * x.F(1, l)); break;
    case OP_v_add_f32: VLOOP x.WF(l, x.F(0, l) + x.F(1, l)); break;
    case OP_v_sub_f32: VLOOP x.WF(l, x.F(0, l) - x.F(1, l)); break;
    case OP_v_subrev_f32: VLOOP x.WF(l, x.F(1, l) - x.F(0, l)); break;
    case OP_v_fma_f32: VLOOP x.WF(l, fmaf(x.F(0, l), x.F(1, l), x.F(2, l))); break;
    case OP_v_fmac_f32: VLOOP x.WF(l, fmaf(x.F(0, l), x.F(1, l), u2f(w.v[(size_t)in.o[0].reg * 64 + l]))); break;
    case OP_v_mac_f32: VLOOP x.WF(l, x.F(0, l) * x.F(1, l) + u2f(w.v[(size_t)in.o[0].reg * 64 + l])); break;
    case OP_v_mad_f32: VLOOP x.WF(l, x.F(0, l) * x.F(1, l) + x.F(2, l)); break;
    case OP_v_max_f32: VLOOP x.WF(l, fmaxf(x.F(0, l), x.F(1, l))); break;
    case OP_v_min_f32: VLOOP x.WF(l, fminf(x.F(0, l), x.F(1, l))); break;
    case OP_v_floor_f32: VLOOP x.WF(l, floorf(x.F(0, l))); break;
    case OP_v_trunc_f32: VLOOP x.WF(l, truncf(x.F(0, l))); break;
    case OP_v_rndne_f32: VLOOP x.WF(l, nearbyintf(x.F(0, l))); break;
    case OP_v_fract_f32: VLOOP { const float f = x.F(0, l); x.WF(l, f - floorf(f)); } break;
    // ---- f64
    case OP_v_add_f64: VLOOP x.WD(l, x.D(0, l) + x.D(1, l)); break;
    case OP_v_mul_f64: VLOOP x.WD(l, x.D(0, l) * x.D(1, l)); break;
    case OP_v_fma_f64: VLOOP x.WD(l, fma(x.D(0, l), x.D(1, l), x.D(2, l))); break;
    case OP_v_fmac_f64: VLOOP { const double acc = u2d((u64)w.v[(size_t)in.o[0].reg * 64 + l] | ((u64)w.v[(size_t)(in.o[0].reg + 1) * 64 + l] << 32)); x.WD(l, fma(x.D(0, l), x.D(1, l), acc)); } break;
    case OP_v_max_f64: VLOOP x.WD(l, fmax(x.D(0, l), x.D(1, l))); break;
    case OP_v_min_f64: VLOOP x.WD(l, fmin(x.D(0, l), x.D(1, l))); break;
    case OP_v_floor_f64: VLOOP x.WD(l, floor(x.D(0, l))); break;
    case OP_v_trunc_f64: VLOOP x.WD(l, trunc(x.D(0, l))); break;
    case OP_v_ceil_f64: VLOOP x.WD(l, ceil(x.D(0, l))); break;
    case OP_v_rndne_f64: VLOOP x.WD(l, nearbyint(x.D(0, l))); break;
    case OP_v_fract_f64: VLOOP { const double f = x.D(0, l); x.WD(l, f - floor(f)); } break;
    case OP_v_cvt_f64_i32: VLOOP x.WD(l, (double)(i32)x.S(0, l)); break;
    case OP_v_cvt_f64_u32: VLOOP x.WD(l, (double)x.S(0, l)); break;
    case OP_v_cvt_i32_f64: VLOOP { const double f = x.D(0, l); x.W(l, f != f ? 0u : f <= -2147483648.0 ? 0x80000000u : f >= 2147483647.0 ? 0x7FFFFFFFu : (u32)(i32)f); } break;
    case OP_v_cvt_u32_f64: VLOOP { const double f = x.D(0, l); x.W(l, f != f ? 0u : f <= 0.0 ? 0u : f >= 4294967295.0 ? 0xFFFFFFFFu : (u32)f); } break;
    case OP_v_ldexp_f64: VLOOP x.WD(l, ldexp(x.D(0, l), (i32)x.S(1, l))); break;
    case OP_v_frexp_mant_f64: VLOOP { const double f = x.D(0, l); int e; x.WD(l, (f == 0 || f != f || isinf(f)) ? f : frexp(f, &e)); } break;
    case OP_v_frexp_exp_i32_f64: VLOOP { const double f = x.D(0, l); int e = 0; if (!(f == 0 || f != f || isinf(f))) frexp(f, &e); x.W(l, (u32)e); } break;
    case OP_v_div_fmas_f64: VLOOP { const double r = fma(x.D(0, l), x.D(1, l), x.D(2, l)); x.WD(l, (w.vcc >> l & 1) ? ldexp(r, 64) : r); } break;
    case OP_v_div_fixup_f64: VLOOP x.WD(l, div_fixup_f64(x.D(0, l), x.D(1, l), x.D(2, l))); break;
    case OP_v_rcp_f64: VLOOP x.WD(l, 1.0 / x.D(0, l)); break;
    case OP_v_rsq_f64: VLOOP x.WD(l, 1.0 / sqrt(x.D(0, l))); break;
    case OP_v_sqrt_f64: VLOOP x.WD(l, sqrt(x.D(0, l))); break;
    default:
        emu_fault(w, "vector instruction not implemented");
        return false;
    }
#undef I1
#undef I2
#undef I3
    return true;
}

// ---------------------------------------------------------------------------------------------------------------- LDS
static bool exec_ds(Wave &w, const Inst &in)
{
    std::vector<u8> &L = w.wg->lds;
    const u16 op = in.op;
    auto V = [&](int k, int l, int dw = 0) -> u32 & { return w.v[(size_t)(in.o[k].reg + dw) * 64 + l]; };
    auto ok = [&](u64 a, u32 n) -> bool { if (a + n > L.size()) { emu_fault(w, "LDS access of %u bytes at %llu beyond %zu", n, (unsigned long long)a, L.size()); return false; } return true; };
#define DLOOP for (int l = 0; l < 64; l++) if (w.exec >> l & 1)
    auto rd = [&](int n, bool sx, int mode) {       // mode 0: whole, 1: d16 low, 2: d16 high
        DLOOP {
            const u64 a = (u64)V(1, l) + (u32)in.off0;
            if (!ok(a, (u32)n)) return false;
            if (n <= 2) {
                u32 v = 0; memcpy(&v, &L[a], n);
                if (sx) v = n == 1 ? (u32)(i32)(i8)v : (u32)(i32)(i16)v;
                if (mode == 1) V(0, l) = (V(0, l) & 0xFFFF0000u) | (v & 0xFFFF);
                else if (mode == 2) V(0, l) = (V(0, l) & 0xFFFFu) | (v << 16);
                else V(0, l) = v;
            }
            else { u32 t[4]; memcpy(t, &L[a], n); for (int k = 0; k < n / 4; k++) V(0, l, k) = t[k]; }
        }
        return true;
    };
    auto wr = [&](int n, int shift) {
        DLOOP {
            const u64 a = (u64)V(0, l) + (u32)in.off0;
            if (!ok(a, (u32)n)) return false;
            if (n <= 2) { const u32 v = V(1, l) >> shift; memcpy(&L[a], &v, n); }
            else { u32 t[4]; for (int k = 0; k < n / 4; k++) t[k] = V(1, l, k); memcpy(&L[a], t, n); }
        }
        return true;
    };
    auto rd2 = [&](int n, int stride) {
        DLOOP {
            const u64 a0 = (u64)V(1, l) + (u64)in.off0 * stride, a1 = (u64)V(1, l) + (u64)in.off1 * stride;
            if (!ok(a0, (u32)n) || !ok(a1, (u32)n)) return false;
            u32 t[4]; memcpy(t, &L[a0], n); memcpy(t + n / 4, &L[a1], n);
            for (int k = 0; k < n / 2; k++) V(0, l, k) = t[k];
        }
        return true;
    };
    auto wr2 = [&](int n, int stride) {
        DLOOP {
            const u64 a0 = (u64)V(0, l) + (u64)in.off0 * stride, a1 = (u64)V(0, l) + (u64)in.off1 * stride;
            if (!ok(a0, (u32)n) || !ok(a1, (u32)n)) return false;
            u32 t0[2], t1[2]; for (int k = 0; k < n / 4; k++) { t0[k] = V(1, l, k); t1[k] = V(2, l, k); }
            memcpy(&L[a0], t0, n); memcpy(&L[a1], t1, n);
        }
        return true;
    };
    auto atom = [&](auto fn, bool rtn) {
        const int ai = rtn ? 1 : 0, di = rtn ? 2 : 1;
        DLOOP {
            const u64 a = (u64)V(ai, l) + (u32)in.off0;
            if (!ok(a, 4)) return false;
            u32 old; memcpy(&old, &L[a], 4);
            const u32 nv = fn(old, V(di, l));
            memcpy(&L[a], &nv, 4);
            if (rtn) V(0, l) = old;
        }
        return true;
    };
    switch (op) {
    case OP_ds_read_b32: return rd(4, false, 0); case OP_ds_read_b64: return rd(8, false, 0); case OP_ds_read_b96: return rd(12, false, 0);
    case OP_ds_read_b128: return rd(16, false, 0); case OP_ds_read_u8: return rd(1, false, 0); case OP_ds_read_i8: return rd(1, true, 0);
    case OP_ds_read_u16: return rd(2, false, 0); case OP_ds_read_i16: return rd(2, true, 0);
    case OP_ds_read_u16_d16: return rd(2, false, 1); case OP_ds_read_u16_d16_hi: return rd(2, false, 2);
    case OP_ds_read2_b32: return rd2(4, 4); case OP_ds_read2_b64: return rd2(8, 8); case OP_ds_read2st64_b32: return rd2(4, 256); case OP_ds_read2st64_b64: return rd2(8, 512);
    case OP_ds_write_b8: return wr(1, 0); case OP_ds_write_b16: return wr(2, 0); case OP_ds_write_b32: return wr(4, 0); case OP_ds_write_b64: return wr(8, 0);
    case OP_ds_write_b96: return wr(12, 0); case OP_ds_write_b128: return wr(16, 0); case OP_ds_write_b8_d16_hi: return wr(1, 16); case OP_ds_write_b16_d16_hi: return wr(2, 16);
    case OP_ds_write2_b32: return wr2(4, 4); case OP_ds_write2_b64: return wr2(8, 8); case OP_ds_write2st64_b32: return wr2(4, 256); case OP_ds_write2st64_b64: return wr2(8, 512);
    case OP_ds_or_b32: return atom([](u32 o, u32 d) { return o | d; }, false);
    case OP_ds_xor_b32: return atom([](u32 o, u32 d) { return o ^ d; }, false);
    case OP_ds_and_b32: return atom([](u32 o, u32 d) { return o & d; }, false);
    case OP_ds_add_u32: return atom([](u32 o, u32 d) { return o + d; }, false);
    case OP_ds_sub_u32: return atom([](u32 o, u32 d) { return o - d; }, false);
    case OP_ds_max_u32: return atom([](u32 o, u32 d) { return o > d ? o : d; }, false);
    case OP_ds_min_u32: return atom([](u32 o, u32 d) { return o < d ? o : d; }, false);
    case OP_ds_max_i32: return atom([](u32 o, u32 d) { return (i32)o > (i32)d ? o : d; }, false);
    case OP_ds_min_i32: return atom([](u32 o, u32 d) { return (i32)o < (i32)d ? o : d; }, false);
    case OP_ds_inc_u32: return atom([](u32 o, u32 d) { return o >= d ? 0u : o + 1; }, false);
    case OP_ds_add_rtn_u32: return atom([](u32 o, u32 d) { return o + d; }, true);
    case OP_ds_or_rtn_b32: return atom([](u32 o, u32 d) { return o | d; }, true);
    case OP_ds_max_rtn_u32: return atom([](u32 o, u32 d) { return o > d ? o : d; }, true);
    case OP_ds_min_rtn_u32: return atom([](u32 o, u32 d) { return o < d ? o : d; }, true);
    case OP_ds_inc_rtn_u32: return atom([](u32 o, u32 d) { return o >= d ? 0u : o + 1; }, true);
    case OP_ds_wrxchg_rtn_b32: return atom([](u32, u32 d) { return d; }, true);
    case OP_ds_bpermute_b32: {
        u32 src[64], out[64];
        for (int l = 0; l < 64; l++) src[l] = (w.exec >> l & 1) ? V(2, l) : 0u;       // a lane that is off gives 0
        DLOOP out[l] = src[((V(1, l) + (u32)in.off0) >> 2) & 63];
        DLOOP V(0, l) = out[l];
        return true;
    }
    case OP_ds_permute_b32: {
        u32 out[64] = {0};
        DLOOP out[((V(1, l) + (u32)in.off0) >> 2) & 63] = V(2, l);
        DLOOP V(0, l) = out[l];
        return true;
    }
    case OP_ds_nop: return true;
    default: emu_fault(w, "LDS instruction not implemented"); return false;
    }
#undef DLOOP
}

// ---------------------------------------------------------------------------------------------------------------- global / flat / scratch
static bool exec_mem(Wave &w, const Inst &in)
{
    const u16 op = in.op;
    const char *name = op_name(op);
    const bool is_global = !strncmp(name, "global_", 7), is_flat = !strncmp(name, "flat_", 5), is_scratch = !strncmp(name, "scratch_", 8);
    const char *what = name + (is_global ? 7 : is_flat ? 5 : 8);
    const bool is_load = !strncmp(what, "load_", 5), is_store = !strncmp(what, "store_", 6), is_atomic = !strncmp(what, "atomic_", 7);
    const bool to_lds = is_load && !strncmp(what, "load_lds_", 9);
    KStats *st = w.d->stats;
    auto V = [&](int k, int l, int dw = 0) -> u32 & { return w.v[(size_t)(in.o[k].reg + dw) * 64 + l]; };
    // operand positions
    int vdst = -1, vaddr, vdata = -1, saddr = -1;
    if (to_lds) { vaddr = 0; saddr = 1; }
    else if (is_load) { vdst = 0; vaddr = 1; saddr = 2; }
    else if (is_store) { vaddr = 0; vdata = 1; saddr = 2; }
    else { if (in.ret) { vdst = 0; vaddr = 1; vdata = 2; saddr = 3; } else { vaddr = 0; vdata = 1; saddr = 2; } }
    if (is_flat) saddr = -1;
    auto address = [&](int l, u32 n, u8 **p) -> bool {
        if (is_scratch) {
            // scratch_* vdst/vdata, vaddr (off or a VGPR), saddr (off or an SGPR): per-lane private memory
            u64 a = (u64)(i64)in.off0;
            if (in.o[vaddr].kind == K_VGPR) a += V(vaddr, l);
            if (saddr >= 0 && saddr < in.no && in.o[saddr].kind == K_SGPR) a += w.s[in.o[saddr].reg];
            const u32 sz = w.d->ki->scratch;
            if (a + n > sz) { emu_fault(w, "scratch access of %u bytes at %llu beyond %u", n, (unsigned long long)a, sz); return false; }
            *p = w.scratch.data() + (size_t)l * sz + a;
            return true;
        }
        u64 a;
        if (saddr >= 0 && saddr < in.no && in.o[saddr].kind == K_SGPR) a = rs64(w, in.o[saddr]) + (u64)V(vaddr, l);
        else a = (u64)V(vaddr, l) | ((u64)V(vaddr, l, 1) << 32);
        a += (u64)(i64)in.off0;
        if (is_flat) { *p = flat_ptr(w, a, n, l); return *p != nullptr; }
        if (!glob_ok(w, a, n)) return false;
        *p = (u8 *)(uintptr_t)a;
        return true;
    };
#define MLOOP for (int l = 0; l < 64; l++) if (w.exec >> l & 1)
    if (to_lds) {
        const u32 n = op == OP_global_load_lds_dwordx4 ? 16 : op == OP_global_load_lds_dwordx3 ? 12 : op == OP_global_load_lds_dword ? 4 : op == OP_global_load_lds_ushort ? 2 : 1;
        const u32 unit = n < 4 ? 4 : n;
        MLOOP {
            u8 *p; if (!address(l, n, &p)) return false;
            const u64 la = (u64)(w.m0 & 0xFFFFF) + (u32)in.off0 + (u64)l * unit;
            if (la + unit > w.wg->lds.size()) { emu_fault(w, "LDS-DMA write at %llu beyond %zu", (unsigned long long)la, w.wg->lds.size()); return false; }
            u8 linebuf[16];
            if (g_emu_l1_cus && !in.sc1 && !in.nt && emu_l1_read(w.wg->cu, (u64)(uintptr_t)p, linebuf, n)) p = linebuf;
            if (n < 4) { u32 v = 0; memcpy(&v, p, n); memcpy(&w.wg->lds[la], &v, 4); } else memcpy(&w.wg->lds[la], p, n);
            if (st) st->global_load_bytes += n;
        }
        return true;
    }
    if (is_load) {
        const char *t = what + 5;
        u32 n; bool sx = false; int mode = 0;      // mode 1: d16 (low half kept... written), 2: d16_hi
        if (!strcmp(t, "ubyte")) n = 1; else if (!strcmp(t, "sbyte")) { n = 1; sx = true; } else if (!strcmp(t, "ushort")) n = 2;
        else if (!strcmp(t, "sshort")) { n = 2; sx = true; } else if (!strcmp(t, "dword")) n = 4; else if (!strcmp(t, "dwordx2")) n = 8;
        else if (!strcmp(t, "dwordx3")) n = 12; else if (!strcmp(t, "dwordx4")) n = 16;
        else if (!strcmp(t, "short_d16")) { n = 2; mode = 1; } else if (!strcmp(t, "short_d16_hi")) { n = 2; mode = 2; }
        else if (!strcmp(t, "ubyte_d16")) { n = 1; mode = 1; } else if (!strcmp(t, "ubyte_d16_hi")) { n = 1; mode = 2; }
        else { emu_fault(w, "load form not implemented"); return false; }
        // (all lanes read before any lane's destination is written: the address registers may be the destination)
        u32 tmp[64][4];
        const bool cached = g_emu_l1_cus && !is_scratch && !((in.sc1 || in.nt) && emu_l1_sc1_bypasses());
        MLOOP {
            u8 *p; if (!address(l, n, &p)) return false;
            tmp[l][0] = 0;
            if (!(cached && (u64)(uintptr_t)p >> 32 != EMU_SHARED_HI && !(is_flat && (p >= w.wg->lds.data() && p < w.wg->lds.data() + w.wg->lds.size())) &&
                  !(is_flat && !w.scratch.empty() && p >= w.scratch.data() && p < w.scratch.data() + w.scratch.size()) &&
                  emu_l1_read(w.wg->cu, (u64)(uintptr_t)p, tmp[l], n)))
                memcpy(tmp[l], p, n);
            if (sx) tmp[l][0] = n == 1 ? (u32)(i32)(i8)tmp[l][0] : (u32)(i32)(i16)tmp[l][0];
            if (st && !is_scratch) st->global_load_bytes += n;
        }
        MLOOP {
            if (mode == 1) V(vdst, l) = (V(vdst, l) & 0xFFFF0000u) | (tmp[l][0] & 0xFFFF);
            else if (mode == 2) V(vdst, l) = (V(vdst, l) & 0xFFFFu) | (tmp[l][0] << 16);
            else for (u32 k = 0; k < (n + 3) / 4; k++) V(vdst, l, (int)k) = tmp[l][k];
        }
        return true;
    }
    if (is_store) {
        const char *t = what + 6;
        u32 n; int shift = 0;
        if (!strcmp(t, "byte")) n = 1; else if (!strcmp(t, "short")) n = 2; else if (!strcmp(t, "dword")) n = 4; else if (!strcmp(t, "dwordx2")) n = 8;
        else if (!strcmp(t, "dwordx3")) n = 12; else if (!strcmp(t, "dwordx4")) n = 16;
        else if (!strcmp(t, "byte_d16_hi")) { n = 1; shift = 16; } else if (!strcmp(t, "short_d16_hi")) { n = 2; shift = 16; }
        else { emu_fault(w, "store form not implemented"); return false; }
        MLOOP {
            u8 *p; if (!address(l, n, &p)) return false;
            u32 t4[4];
            for (u32 k = 0; k < (n + 3) / 4; k++) t4[k] = V(vdata, l, (int)k);
            if (n <= 2) t4[0] >>= shift;
            if (n == 4) __atomic_store_n((u32 *)p, t4[0], __ATOMIC_RELEASE);          // (a word the host polls must arrive whole and after what precedes it)
            else if (n == 8 && ((uintptr_t)p & 7) == 0) { u64 v; memcpy(&v, t4, 8); __atomic_store_n((u64 *)p, v, __ATOMIC_RELEASE); }
            else memcpy(p, t4, n);
            if (g_emu_l1_cus && !is_scratch) emu_l1_store(w.wg->cu, (u64)(uintptr_t)p, p, n);
            if (st && !is_scratch) st->global_store_bytes += n;
        }
        return true;
    }
    if (is_atomic) {
        const char *t = what + 7;
        const bool x2 = strlen(t) > 3 && !strcmp(t + strlen(t) - 3, "_x2");
        std::string kind = x2 ? std::string(t, strlen(t) - 3) : std::string(t);
        const u32 n = x2 ? 8 : 4;
        MLOOP {
            u8 *p; if (!address(l, n, &p)) return false;
            u64 old = 0; memcpy(&old, p, n);
            u64 d = x2 ? ((u64)V(vdata, l) | ((u64)V(vdata, l, 1) << 32)) : (u64)V(vdata, l);
            u64 nv;
            const u64 mask = x2 ? ~0ull : 0xFFFFFFFFull;
            auto sx = [&](u64 v) -> i64 { return x2 ? (i64)v : (i64)(i32)(u32)v; };
            if (kind == "or") nv = old | d; else if (kind == "and") nv = old & d; else if (kind == "xor") nv = old ^ d;
            else if (kind == "add") nv = old + d; else if (kind == "sub") nv = old - d;
            else if (kind == "umin") nv = old < d ? old : d; else if (kind == "umax") nv = old > d ? old : d;
            else if (kind == "smin") nv = sx(old) < sx(d) ? old : d; else if (kind == "smax") nv = sx(old) > sx(d) ? old : d;
            else if (kind == "inc") nv = old >= d ? 0 : old + 1; else if (kind == "dec") nv = (old == 0 || old > d) ? d : old - 1;
            else if (kind == "swap") nv = d;
            else if (kind == "cmpswap") {
                const u64 cmp = x2 ? ((u64)V(vdata, l, 2) | ((u64)V(vdata, l, 3) << 32)) : (u64)V(vdata, l, 1);
                nv = old == cmp ? d : old;
            }
            else { emu_fault(w, "atomic not implemented"); return false; }
            nv &= mask;
            if (n == 4) __atomic_store_n((u32 *)p, (u32)nv, __ATOMIC_SEQ_CST); else __atomic_store_n((u64 *)p, nv, __ATOMIC_SEQ_CST);
            if (g_emu_l1_cus) emu_l1_drop_line(w.wg->cu, (u64)(uintptr_t)p, n);        // (atomics are done past the L1)
            if (in.ret) { V(vdst, l) = (u32)old; if (x2) V(vdst, l, 1) = (u32)(old >> 32); }
        }
        return true;
    }
#undef MLOOP
    emu_fault(w, "memory instruction not implemented");
    return false;
}

// ---------------------------------------------------------------------------------------------------------------- scalar
static bool exec_salu(Wave &w, const Inst &in, bool *jumped)
{
    const u16 op = in.op;
    const Opnd *o = in.o;
    CodeObject *co = w.d->co;
    auto pc_addr = [&](u32 idx) -> u64 { return (u64)(uintptr_t)co->image.data() + co->insts[idx].addr; };
    auto jump_addr = [&](u64 a) -> bool {
        const u64 rel = a - (u64)(uintptr_t)co->image.data();
        auto it = co->at.find(rel);
        if (it == co->at.end()) { emu_fault(w, "jump to 0x%llx: no instruction there", (unsigned long long)rel); return false; }
        w.pc = it->second; *jumped = true; return true;
    };
#define B32(NAME, EXPR) case OP_##NAME: { const u32 a = rs32(w, o[1]), b = rs32(w, o[2]); (void)b; const u32 r = (EXPR); ws32(w, o[0], r); w.scc = r != 0; } break;
#define B64(NAME, EXPR) case OP_##NAME: { const u64 a = rs64(w, o[1]), b = rs64(w, o[2]); (void)b; const u64 r = (EXPR); ws64(w, o[0], r); w.scc = r != 0; } break;
#define SAVEEXEC(NAME, EXPR) case OP_##NAME: { const u64 a = rs64(w, o[1]), e = w.exec; ws64(w, o[0], e); w.exec = (EXPR); w.scc = w.exec != 0; } break;
#define CMP(NAME, T, OPR) case OP_s_cmp_##NAME: w.scc = (T)rs32(w, o[0]) OPR (T)rs32(w, o[1]); break;
#define CMPK(NAME, T, EXT, OPR) case OP_s_cmpk_##NAME: w.scc = (T)rs32(w, o[0]) OPR (T)(EXT)(u16)o[1].imm; break;
    switch (op) {
    case OP_s_mov_b32: ws32(w, o[0], rs32(w, o[1])); break;
    case OP_s_mov_b64: ws64(w, o[0], rs64(w, o[1])); break;
    case OP_s_movk_i32: ws32(w, o[0], (u32)(i32)(i16)(u16)o[1].imm); break;
    B32(s_and_b32, a & b) B32(s_or_b32, a | b) B32(s_xor_b32, a ^ b) B32(s_andn2_b32, a & ~b) B32(s_orn2_b32, a | ~b) B32(s_nor_b32, ~(a | b)) B32(s_nand_b32, ~(a & b)) B32(s_xnor_b32, ~(a ^ b))
    B64(s_and_b64, a & b) B64(s_or_b64, a | b) B64(s_xor_b64, a ^ b) B64(s_andn2_b64, a & ~b) B64(s_orn2_b64, a | ~b) B64(s_nor_b64, ~(a | b)) B64(s_nand_b64, ~(a & b)) B64(s_xnor_b64, ~(a ^ b))
    case OP_s_not_b32: { const u32 r = ~rs32(w, o[1]); ws32(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_not_b64: { const u64 r = ~rs64(w, o[1]); ws64(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_add_i32: { const i32 a = (i32)rs32(w, o[1]), b = (i32)rs32(w, o[2]); i32 r; w.scc = __builtin_add_overflow(a, b, &r); ws32(w, o[0], (u32)r); } break;
    case OP_s_sub_i32: { const i32 a = (i32)rs32(w, o[1]), b = (i32)rs32(w, o[2]); i32 r; w.scc = __builtin_sub_overflow(a, b, &r); ws32(w, o[0], (u32)r); } break;
    case OP_s_add_u32: { const u64 r = (u64)rs32(w, o[1]) + rs32(w, o[2]); ws32(w, o[0], (u32)r); w.scc = r >> 32; } break;
    case OP_s_addc_u32: { const u64 r = (u64)rs32(w, o[1]) + rs32(w, o[2]) + (w.scc ? 1 : 0); ws32(w, o[0], (u32)r); w.scc = r >> 32; } break;
    case OP_s_sub_u32: { const u32 a = rs32(w, o[1]), b = rs32(w, o[2]); ws32(w, o[0], a - b); w.scc = b > a; } break;
    case OP_s_subb_u32: { const u64 a = rs32(w, o[1]), b = (u64)rs32(w, o[2]) + (w.scc ? 1 : 0); ws32(w, o[0], (u32)(a - b)); w.scc = b > a; } break;
    case OP_s_mul_i32: ws32(w, o[0], rs32(w, o[1]) * rs32(w, o[2])); break;
    case OP_s_mul_hi_u32: ws32(w, o[0], (u32)(((u64)rs32(w, o[1]) * rs32(w, o[2])) >> 32)); break;
    case OP_s_mul_hi_i32: ws32(w, o[0], (u32)((u64)((i64)(i32)rs32(w, o[1]) * (i32)rs32(w, o[2])) >> 32)); break;
    B32(s_lshl_b32, a << (b & 31)) B32(s_lshr_b32, a >> (b & 31)) B32(s_ashr_i32, (u32)((i32)a >> (b & 31)))
    case OP_s_lshl_b64: { const u64 r = rs64(w, o[1]) << (rs32(w, o[2]) & 63); ws64(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_lshr_b64: { const u64 r = rs64(w, o[1]) >> (rs32(w, o[2]) & 63); ws64(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_ashr_i64: { const u64 r = (u64)((i64)rs64(w, o[1]) >> (rs32(w, o[2]) & 63)); ws64(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_bfe_u32: { const u32 a = rs32(w, o[1]), b = rs32(w, o[2]), off = b & 31, wd = (b >> 16) & 0x7F; const u32 r = wd == 0 ? 0 : wd >= 32 ? a >> off : (a >> off) & ((1u << wd) - 1); ws32(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_bfe_i32: { const u32 a = rs32(w, o[1]), b = rs32(w, o[2]), off = b & 31, wd = (b >> 16) & 0x7F; u32 r = 0;
                         if (wd >= 32) r = (u32)((i32)a >> off); else if (wd) { const u32 f = (a >> off) & ((1u << wd) - 1); r = (u32)((i32)(f << (32 - wd)) >> (32 - wd)); }
                         ws32(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_bfe_u64: { const u64 a = rs64(w, o[1]); const u32 b = rs32(w, o[2]), off = b & 63, wd = (b >> 16) & 0x7F; const u64 r = wd == 0 ? 0 : wd >= 64 ? a >> off : (a >> off) & ((1ull << wd) - 1); ws64(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_bfm_b32: ws32(w, o[0], ((1u << (rs32(w, o[1]) & 31)) - 1u) << (rs32(w, o[2]) & 31)); break;
    case OP_s_bfm_b64: ws64(w, o[0], ((1ull << (rs32(w, o[1]) & 63)) - 1ull) << (rs32(w, o[2]) & 63)); break;
    case OP_s_min_u32: { const u32 a = rs32(w, o[1]), b = rs32(w, o[2]); w.scc = a < b; ws32(w, o[0], a < b ? a : b); } break;
    case OP_s_max_u32: { const u32 a = rs32(w, o[1]), b = rs32(w, o[2]); w.scc = a > b; ws32(w, o[0], a > b ? a : b); } break;
    case OP_s_min_i32: { const i32 a = (i32)rs32(w, o[1]), b = (i32)rs32(w, o[2]); w.scc = a < b; ws32(w, o[0], (u32)(a < b ? a : b)); } break;
    case OP_s_max_i32: { const i32 a = (i32)rs32(w, o[1]), b = (i32)rs32(w, o[2]); w.scc = a > b; ws32(w, o[0], (u32)(a > b ? a : b)); } break;
    case OP_s_cselect_b32: ws32(w, o[0], w.scc ? rs32(w, o[1]) : rs32(w, o[2])); break;
    case OP_s_cselect_b64: ws64(w, o[0], w.scc ? rs64(w, o[1]) : rs64(w, o[2])); break;
    case OP_s_abs_i32: { const i32 a = (i32)rs32(w, o[1]); const u32 r = (u32)(a < 0 ? -(i64)a : a); ws32(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_sext_i32_i8: ws32(w, o[0], (u32)(i32)(i8)rs32(w, o[1])); break;
    case OP_s_sext_i32_i16: ws32(w, o[0], (u32)(i32)(i16)rs32(w, o[1])); break;
    case OP_s_lshl1_add_u32: case OP_s_lshl2_add_u32: case OP_s_lshl3_add_u32: case OP_s_lshl4_add_u32: {
        const int sh = op == OP_s_lshl1_add_u32 ? 1 : op == OP_s_lshl2_add_u32 ? 2 : op == OP_s_lshl3_add_u32 ? 3 : 4;
        const u64 r = ((u64)rs32(w, o[1]) << sh) + rs32(w, o[2]); ws32(w, o[0], (u32)r); w.scc = r >> 32 != 0; } break;
    case OP_s_pack_ll_b32_b16: ws32(w, o[0], (rs32(w, o[1]) & 0xFFFF) | (rs32(w, o[2]) << 16)); break;
    SAVEEXEC(s_and_saveexec_b64, a & e) SAVEEXEC(s_or_saveexec_b64, a | e) SAVEEXEC(s_andn2_saveexec_b64, a & ~e) SAVEEXEC(s_xor_saveexec_b64, a ^ e)
    SAVEEXEC(s_orn2_saveexec_b64, a | ~e) SAVEEXEC(s_andn1_saveexec_b64, ~a & e)
    case OP_s_brev_b32: ws32(w, o[0], brev32(rs32(w, o[1]))); break;
    case OP_s_brev_b64: ws64(w, o[0], brev64(rs64(w, o[1]))); break;
    case OP_s_bcnt1_i32_b32: { const u32 r = (u32)__builtin_popcount(rs32(w, o[1])); ws32(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_bcnt1_i32_b64: { const u32 r = (u32)__builtin_popcountll(rs64(w, o[1])); ws32(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_bcnt0_i32_b32: { const u32 r = 32u - (u32)__builtin_popcount(rs32(w, o[1])); ws32(w, o[0], r); w.scc = r != 0; } break;
    case OP_s_ff1_i32_b32: { const u32 a = rs32(w, o[1]); ws32(w, o[0], a ? (u32)__builtin_ctz(a) : 0xFFFFFFFFu); } break;
    case OP_s_ff1_i32_b64: { const u64 a = rs64(w, o[1]); ws32(w, o[0], a ? (u32)__builtin_ctzll(a) : 0xFFFFFFFFu); } break;
    case OP_s_ff0_i32_b32: { const u32 a = ~rs32(w, o[1]); ws32(w, o[0], a ? (u32)__builtin_ctz(a) : 0xFFFFFFFFu); } break;
    case OP_s_flbit_i32_b32: { const u32 a = rs32(w, o[1]); ws32(w, o[0], a ? (u32)__builtin_clz(a) : 0xFFFFFFFFu); } break;
    case OP_s_flbit_i32_b64: { const u64 a = rs64(w, o[1]); ws32(w, o[0], a ? (u32)__builtin_clzll(a) : 0xFFFFFFFFu); } break;
    case OP_s_flbit_i32: { const u32 a = rs32(w, o[1]); ws32(w, o[0], (a == 0 || a == 0xFFFFFFFFu) ? 0xFFFFFFFFu : (u32)__builtin_clz((i32)a < 0 ? ~a : a)); } break;
    case OP_s_bitset1_b32: ws32(w, o[0], rs32(w, o[0]) | (1u << (rs32(w, o[1]) & 31))); break;
    case OP_s_bitset0_b32: ws32(w, o[0], rs32(w, o[0]) & ~(1u << (rs32(w, o[1]) & 31))); break;
    case OP_s_bitset1_b64: ws64(w, o[0], rs64(w, o[0]) | (1ull << (rs32(w, o[1]) & 63))); break;
    case OP_s_bitset0_b64: ws64(w, o[0], rs64(w, o[0]) & ~(1ull << (rs32(w, o[1]) & 63))); break;
    case OP_s_bitcmp0_b32: w.scc = !(rs32(w, o[0]) >> (rs32(w, o[1]) & 31) & 1); break;
    case OP_s_bitcmp1_b32: w.scc = (rs32(w, o[0]) >> (rs32(w, o[1]) & 31) & 1); break;
    case OP_s_bitcmp0_b64: w.scc = !(rs64(w, o[0]) >> (rs32(w, o[1]) & 63) & 1); break;
    case OP_s_bitcmp1_b64: w.scc = (rs64(w, o[0]) >> (rs32(w, o[1]) & 63) & 1); break;
    CMP(eq_i32, i32, ==) CMP(lg_i32, i32, !=) CMP(gt_i32, i32, >) CMP(ge_i32, i32, >=) CMP(lt_i32, i32, <) CMP(le_i32, i32, <=)
    CMP(eq_u32, u32, ==) CMP(lg_u32, u32, !=) CMP(gt_u32, u32, >) CMP(ge_u32, u32, >=) CMP(lt_u32, u32, <) CMP(le_u32, u32, <=)
    case OP_s_cmp_eq_u64: w.scc = rs64(w, o[0]) == rs64(w, o[1]); break;
    case OP_s_cmp_lg_u64: w.scc = rs64(w, o[0]) != rs64(w, o[1]); break;
    CMPK(eq_i32, i32, i16, ==) CMPK(lg_i32, i32, i16, !=) CMPK(gt_i32, i32, i16, >) CMPK(ge_i32, i32, i16, >=) CMPK(lt_i32, i32, i16, <) CMPK(le_i32, i32, i16, <=)
    CMPK(eq_u32, u32, u16, ==) CMPK(lg_u32, u32, u16, !=) CMPK(gt_u32, u32, u16, >) CMPK(ge_u32, u32, u16, >=) CMPK(lt_u32, u32, u16, <) CMPK(le_u32, u32, u16, <=)
    case OP_s_addk_i32: { const i32 a = (i32)rs32(w, o[0]), b = (i32)(i16)(u16)o[1].imm; i32 r; w.scc = __builtin_add_overflow(a, b, &r); ws32(w, o[0], (u32)r); } break;
    case OP_s_mulk_i32: ws32(w, o[0], (u32)((i32)rs32(w, o[0]) * (i32)(i16)(u16)o[1].imm)); break;
    case OP_s_getpc_b64: ws64(w, o[0], pc_addr(w.pc + 1)); break;
    case OP_s_setpc_b64: return jump_addr(rs64(w, o[0]));
    case OP_s_swappc_b64: { const u64 t = rs64(w, o[1]); ws64(w, o[0], pc_addr(w.pc + 1)); return jump_addr(t); }
    case OP_s_set_gpr_idx_on: w.gpr_idx_mode = in.gpr_idx_mode; w.gpr_idx = rs32(w, o[0]) & 0xFF; break;
    case OP_s_set_gpr_idx_off: w.gpr_idx_mode = 0; break;
    case OP_s_branch: w.pc = in.target; *jumped = true; break;
    case OP_s_cbranch_scc0: if (!w.scc) { w.pc = in.target; *jumped = true; } break;
    case OP_s_cbranch_scc1: if (w.scc) { w.pc = in.target; *jumped = true; } break;
    case OP_s_cbranch_vccz: if (w.vcc == 0) { w.pc = in.target; *jumped = true; } break;
    case OP_s_cbranch_vccnz: if (w.vcc != 0) { w.pc = in.target; *jumped = true; } break;
    case OP_s_cbranch_execz: if (w.exec == 0) { w.pc = in.target; *jumped = true; } break;
    case OP_s_cbranch_execnz: if (w.exec != 0) { w.pc = in.target; *jumped = true; } break;
    case OP_s_memrealtime: ws64(w, o[0], g_emu_clock.load(std::memory_order_relaxed)); break;
    case OP_s_memtime: ws64(w, o[0], g_emu_clock.load(std::memory_order_relaxed) * 21); break;
    case OP_s_load_dword: case OP_s_load_dwordx2: case OP_s_load_dwordx4: case OP_s_load_dwordx8: case OP_s_load_dwordx16: {
        const u32 n = op == OP_s_load_dword ? 1 : op == OP_s_load_dwordx2 ? 2 : op == OP_s_load_dwordx4 ? 4 : op == OP_s_load_dwordx8 ? 8 : 16;
        u64 a = rs64(w, o[1]);
        if (in.no >= 3) a += (o[2].kind == K_IMM) ? (u64)o[2].imm : (u64)rs32(w, o[2]);
        a += (u64)(i64)in.off0;
        if (!glob_ok(w, a, 4 * n)) return false;
        u32 t[16]; memcpy(t, (const void *)(uintptr_t)a, 4 * n);
        if (o[0].kind == K_SGPR) for (u32 k = 0; k < n; k++) w.s[o[0].reg + k] = t[k];
        else if (o[0].kind == K_VCC) w.vcc = (u64)t[0] | ((u64)t[1] << 32);
        else ws32(w, o[0], t[0]);
    } break;
    case OP_s_waitcnt: case OP_s_nop: case OP_s_setprio: case OP_s_dcache_wb: case OP_s_dcache_inv: case OP_s_icache_inv: case OP_s_sethalt: break;
    case OP_s_trap: emu_fault(w, "s_trap (an assertion or an abort() in device code)"); return false;
    default: emu_fault(w, "scalar instruction not implemented"); return false;
    }
    return true;
}

bool emu_step(Wave &w)
{
    if (w.state != W_RUN) return false;
    const Inst &in = w.d->co->insts[w.pc];
    KStats *st = w.d->stats;
    const char *name = op_name(in.op);
    if (st) { st->wave_insts++; if (++w.insts > st->max_wave_insts) st->max_wave_insts = w.insts; }
    if (w.d->pc_hist) (*w.d->pc_hist)[w.pc]++;
    if (in.op == OP_s_endpgm || in.op == OP_s_code_end) { w.state = W_DONE; return false; }
    if (in.op == OP_s_barrier) {
        w.pc++;
        if (w.wg->nwaves - w.wg->done > 1) { w.state = W_BARRIER; w.wg->at_barrier++; return false; }
        return true;
    }
    if (in.op == OP_s_sleep) { w.pc++; return false; }          // (gives the turn away; the wave stays runnable)
    bool ok, jumped = false;
    if (name[0] == 's' && name[1] == '_') { ok = exec_salu(w, in, &jumped); if (st) { if (in.op >= OP_s_load_dword && in.op <= OP_s_load_dwordx16) st->smem++; else st->salu++; } }
    else if (name[0] == 'v') {
        ok = exec_valu(w, in);
        if (st) { if (in.op == OP_v_mfma_f64_4x4x4_4b_f64) st->mfma++; st->valu++; st->valu_lanes += (u64)__builtin_popcountll(w.exec); }
    }
    else if (name[0] == 'd') { ok = exec_ds(w, in); if (st) st->lds++; }
    else if (name[0] == 'b') { ok = true; if (g_emu_l1_cus && in.op == OP_buffer_inv && in.sc1) emu_l1_invalidate(w.wg->cu); }     // (buffer_wbl2: stores are write-through here)
    else { ok = exec_mem(w, in); if (st) st->vmem++; }
    if (!ok || w.state == W_FAULT) { w.state = W_FAULT; return false; }
    if (w.trace_lane >= 0) {
        // (GFX950EMU_WATCH: one line per instruction of the watched wave -- what it wrote, for one lane)
        char b[256]; int n = snprintf(b, sizeof b, "T %6u %-28s exec %016llx vcc %016llx scc %d |", in.line, name, (unsigned long long)w.exec, (unsigned long long)w.vcc, (int)w.scc);
        if (in.no > 0) {
            const Opnd &d0 = in.o[0];
            if (d0.kind == K_VGPR) for (int k = 0; k < d0.n && k < 4; k++) n += snprintf(b + n, sizeof b - n, " v%u=%08x", d0.reg + k, w.v[(size_t)(d0.reg + k) * 64 + w.trace_lane]);
            else if (d0.kind == K_SGPR) for (int k = 0; k < d0.n && k < 4; k++) n += snprintf(b + n, sizeof b - n, " s%u=%08x", d0.reg + k, w.s[d0.reg + k]);
        }
        fprintf(stderr, "%s\n", b);
    }
    if (!jumped) w.pc++;
    return true;
}
