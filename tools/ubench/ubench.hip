// Instruction-cost microbenchmarks for a lone wave on a gfx950 SIMD (tuning aid; not part of the product).
// Each test runs REP copies of an instruction pattern inside one wave and reports cycles (s_memtime) per copy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define STR2(x) #x
#define STR(x) STR2(x)
#define REP 256

#define BENCH_BEGIN(name) \
    __global__ void __launch_bounds__(64) name(uint64_t *out, uint32_t *buf, uint32_t seed) { \
        __shared__ uint32_t lds[4096]; \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x55, d = threadIdx.x * 4; \
        double fa = seed * 0.5, fb = 1.0000001; \
        lds[threadIdx.x] = threadIdx.x * 4; lds[threadIdx.x + 64] = a; \
        __syncthreads(); \
        uint64_t t0 = clock64();
#define BENCH_END \
        uint64_t t1 = clock64(); \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0; \
        buf[threadIdx.x] = a + b + c + d + (uint32_t)fa + lds[(a & 63)]; \
    }

BENCH_BEGIN(k_empty)
BENCH_END

BENCH_BEGIN(k_dep_add)
    asm volatile(".rept " STR(REP) "\n v_add_u32 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
BENCH_END

BENCH_BEGIN(k_indep_add)
    asm volatile(".rept " STR(REP) "\n v_add_u32 %0, %2, %3\n v_add_u32 %1, %2, %3\n .endr" : "+v"(a), "+v"(c) : "v"(b), "v"(d));
BENCH_END

BENCH_BEGIN(k_dep_alignbit_ffbh)
    asm volatile(".rept " STR(REP) "\n v_alignbit_b32 %0, %0, %1, %2\n v_ffbh_u32 %0, %0\n .endr" : "+v"(a) : "v"(b), "v"(c));
BENCH_END

BENCH_BEGIN(k_dep_mad24)
    asm volatile(".rept " STR(REP) "\n v_mad_i32_i24 %0, %1, %2, %0\n .endr" : "+v"(a) : "v"(b), "v"(c));
BENCH_END


BENCH_BEGIN(k_indep_mad24)
    asm volatile(".rept " STR(REP) "\n v_mad_i32_i24 %0, %2, %3, %0\n v_mad_i32_i24 %1, %2, %3, %1\n .endr" : "+v"(a), "+v"(d) : "v"(b), "v"(c));
BENCH_END

BENCH_BEGIN(k_dep_mad_i32_i16)
    asm volatile(".rept " STR(REP) "\n v_mad_i32_i16 %0, %1, %2, %0\n .endr" : "+v"(a) : "v"(b), "v"(c));
BENCH_END

BENCH_BEGIN(k_dep_dot2_i32_i16)
    asm volatile(".rept " STR(REP) "\n v_dot2_i32_i16 %0, %1, %2, %0\n .endr" : "+v"(a) : "v"(b), "v"(c));
BENCH_END

BENCH_BEGIN(k_dep_fma_f32)
    float fx = (float)a;
    asm volatile(".rept " STR(REP) "\n v_fma_f32 %0, %1, %2, %0\n .endr" : "+v"(fx) : "v"(b), "v"(c));
    a = (uint32_t)fx;
BENCH_END

BENCH_BEGIN(k_recur8)
    // one sample of the decoder's 8-tap recurrence: mul + 7 mad (24-bit) + shift + add
    asm volatile(".rept " STR(REP) "\n v_mul_i32_i24 %1, %2, %0\n v_mad_i32_i24 %1, %3, %0, %1\n v_mad_i32_i24 %1, %2, %0, %1\n v_mad_i32_i24 %1, %3, %0, %1\n"
                 " v_mad_i32_i24 %1, %2, %0, %1\n v_mad_i32_i24 %1, %3, %0, %1\n v_mad_i32_i24 %1, %2, %0, %1\n v_mad_i32_i24 %1, %3, %0, %1\n"
                 " v_ashrrev_i32 %1, 14, %1\n v_add_u32 %0, %0, %1\n .endr" : "+v"(a), "+v"(d) : "v"(b), "v"(c));
BENCH_END

BENCH_BEGIN(k_recur8_dot2)
    // the same with v_dot2_i32_i16: 4 dot2 + shift + add
    asm volatile(".rept " STR(REP) "\n v_dot2_i32_i16 %1, %2, %0, 0\n v_dot2_i32_i16 %1, %3, %0, %1\n v_dot2_i32_i16 %1, %2, %0, %1\n v_dot2_i32_i16 %1, %3, %0, %1\n"
                 " v_ashrrev_i32 %1, 14, %1\n v_add_u32 %0, %0, %1\n .endr" : "+v"(a), "+v"(d) : "v"(b), "v"(c));
BENCH_END

BENCH_BEGIN(k_dep_mul_lo)
    asm volatile(".rept " STR(REP) "\n v_mul_lo_u32 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
BENCH_END

BENCH_BEGIN(k_dep_mad_u64_u32)
    uint64_t acc = a;
    asm volatile(".rept " STR(REP) "\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n .endr" : "+v"(acc) : "v"(b), "v"(c) : "vcc");
    a = (uint32_t)acc;
BENCH_END

BENCH_BEGIN(k_dep_lshl64)
    uint64_t acc = a;
    asm volatile(".rept " STR(REP) "\n v_lshlrev_b64 %0, %1, %0\n .endr" : "+v"(acc) : "v"(b));
    a = (uint32_t)acc;
BENCH_END

BENCH_BEGIN(k_dep_fma64)
    asm volatile(".rept " STR(REP) "\n v_fma_f64 %0, %1, %1, %0\n .endr" : "+v"(fa) : "v"(fb));
BENCH_END

BENCH_BEGIN(k_indep_fma64)
    double fc = fa + 1.0;
    asm volatile(".rept " STR(REP) "\n v_fma_f64 %0, %2, %2, %0\n v_fma_f64 %1, %2, %2, %1\n .endr" : "+v"(fa), "+v"(fc) : "v"(fb));
    fa += fc;
BENCH_END

BENCH_BEGIN(k_cmp_cndmask)
    asm volatile(".rept " STR(REP) "\n v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %0, vcc\n .endr" : "+v"(a) : "v"(b), "v"(c) : "vcc");
BENCH_END

BENCH_BEGIN(k_cmp_saveexec)
    asm volatile(".rept " STR(REP) "\n v_cmp_lt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n .endr" : "+v"(a) : "v"(b), "v"(c) : "vcc", "s20", "s21");
BENCH_END

BENCH_BEGIN(k_cmp_saveexec_branch)
    asm volatile(".rept " STR(REP) "\n v_cmp_lt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1\n v_add_u32 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n .endr" : "+v"(a) : "v"(b), "v"(c) : "vcc", "s20", "s21");
BENCH_END

BENCH_BEGIN(k_salu_dep)
    uint32_t s = seed;
    asm volatile(".rept " STR(REP) "\n s_add_u32 %0, %0, 3\n .endr" : "+s"(s) : : "scc");
    a += s;
BENCH_END

BENCH_BEGIN(k_valu_salu_mix)
    uint32_t s = seed;
    asm volatile(".rept " STR(REP) "\n v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 3\n .endr" : "+v"(a), "+s"(s) : "v"(b) : "scc");
    a += s;
BENCH_END

BENCH_BEGIN(k_readlane_use)
    uint32_t s = 0;
    asm volatile(".rept " STR(REP) "\n v_readlane_b32 %1, %0, 3\n s_nop 3\n v_add_u32 %0, %0, %1\n .endr" : "+v"(a), "+s"(s));
BENCH_END

BENCH_BEGIN(k_lds_read_dep)
    asm volatile(".rept " STR(REP) "\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(d));
BENCH_END

BENCH_BEGIN(k_lds_write_wait)
    asm volatile(".rept " STR(REP) "\n ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n .endr" : : "v"(d), "v"(a) : "memory");
BENCH_END

BENCH_BEGIN(k_lds_write_nowait)
    asm volatile(".rept " STR(REP) "\n ds_write_b32 %0, %1\n .endr\n s_waitcnt lgkmcnt(0)" : : "v"(d), "v"(a) : "memory");
BENCH_END

BENCH_BEGIN(k_lds_write_b128_nowait)
    uint32_t d16 = threadIdx.x * 16;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 val = {a, b, c, d};
    asm volatile(".rept " STR(REP) "\n ds_write_b128 %0, %1\n .endr\n s_waitcnt lgkmcnt(0)" : : "v"(d16), "v"(val) : "memory");
BENCH_END

BENCH_BEGIN(k_lds_or_nowait)
    asm volatile(".rept " STR(REP) "\n ds_or_b32 %0, %1\n .endr\n s_waitcnt lgkmcnt(0)" : : "v"(d), "v"(a) : "memory");
BENCH_END

BENCH_BEGIN(k_lds_or_pair_scattered)
    uint32_t d2 = ((threadIdx.x * 97) & 1023) * 4;
    asm volatile(".rept " STR(REP) "\n ds_or_b32 %0, %1\n ds_or_b32 %0, %1 offset:4\n .endr\n s_waitcnt lgkmcnt(0)" : : "v"(d2), "v"(a) : "memory");
BENCH_END

BENCH_BEGIN(k_lds_read_u16_nowait)
    uint32_t d2 = threadIdx.x * 136, t;
    asm volatile(".rept " STR(REP) "\n ds_read_u16 %0, %1\n .endr\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(d2) : "memory");
    a += t;
BENCH_END

BENCH_BEGIN(k_writelane_readlane)
    uint32_t s = 5;
    asm volatile(".rept " STR(REP) "\n v_writelane_b32 %0, %1, 3\n v_readlane_b32 %1, %0, 3\n .endr" : "+v"(a), "+s"(s));
    a += s;
BENCH_END

BENCH_BEGIN(k_valu_sgpr_operand)
    uint32_t s = seed;
    asm volatile(".rept " STR(REP) "\n v_mul_i32_i24 %0, %1, %0\n .endr" : "+v"(a) : "s"(s));
BENCH_END

// the per-code step of the decoder's delimiting loop (see flac_dec_fast.hip), as compiled
BENCH_BEGIN(k_rice_step_v1)
    uint32_t w0 = a, w1 = b, w2 = c, w3 = d, sm = 5, smk = 3, lz, p, wb = threadIdx.x * 4, u0, u1;
    asm volatile(".rept " STR(REP) "\n"
        "v_alignbit_b32 %[p], %[w0], %[w1], %[sm]\n"
        "v_ffbh_u32 %[lz], %[p]\n"
        "v_and_b32 %[lz], 15, %[lz]\n"
        "v_cmp_lt_i32 vcc, %[smk], %[lz]\n"
        "s_and_saveexec_b64 s[20:21], vcc\n"
        "s_cbranch_execz 6\n"
        "v_and_b32 %[t0], 0x3ff, %[wb]\n"
        "v_perm_b32 %[t1], 0, %[w3], %[sel]\n"
        "ds_read_b32 %[w3], %[t0]\n"
        "v_add_u32 %[wb], 4, %[wb]\n"
        "v_mov_b32 %[w2], %[t1]\n"
        "s_nop 0\n"
        "s_or_b64 exec, exec, s[20:21]\n"
        "v_sub_u32 %[sm], %[smk], %[lz]\n"
        "v_and_b32 %[sm], 31, %[sm]\n"
        "v_sub_u32 %[smk], %[sm], %[kp1]\n"
        "v_cndmask_b32 %[w0], %[w0], %[w1], vcc\n"
        "v_cndmask_b32 %[w1], %[w1], %[w2], vcc\n"
        ".endr\n s_waitcnt lgkmcnt(0)"
        : [w0]"+v"(w0), [w1]"+v"(w1), [w2]"+v"(w2), [w3]"+v"(w3), [sm]"+v"(sm), [smk]"+v"(smk), [lz]"=&v"(lz), [p]"=&v"(p), [wb]"+v"(wb), [t0]"=&v"(u0), [t1]"=&v"(u1)
        : [sel]"s"(0x00010203u), [kp1]"v"(11u) : "vcc", "s20", "s21", "memory");
    a = w0 + w1 + w2 + w3 + sm;
BENCH_END

// same without the skip branch
BENCH_BEGIN(k_rice_step_v2)
    uint32_t w0 = a, w1 = b, w2 = c, w3 = d, sm = 5, smk = 3, lz, p, wb = threadIdx.x * 4, u0, u1;
    asm volatile(".rept " STR(REP) "\n"
        "v_alignbit_b32 %[p], %[w0], %[w1], %[sm]\n"
        "v_ffbh_u32 %[lz], %[p]\n"
        "v_and_b32 %[lz], 15, %[lz]\n"
        "v_cmp_lt_i32 vcc, %[smk], %[lz]\n"
        "s_and_saveexec_b64 s[20:21], vcc\n"
        "v_and_b32 %[t0], 0x3ff, %[wb]\n"
        "v_perm_b32 %[t1], 0, %[w3], %[sel]\n"
        "ds_read_b32 %[w3], %[t0]\n"
        "v_add_u32 %[wb], 4, %[wb]\n"
        "v_mov_b32 %[w2], %[t1]\n"
        "s_or_b64 exec, exec, s[20:21]\n"
        "v_sub_u32 %[sm], %[smk], %[lz]\n"
        "v_and_b32 %[sm], 31, %[sm]\n"
        "v_sub_u32 %[smk], %[sm], %[kp1]\n"
        "v_cndmask_b32 %[w0], %[w0], %[w1], vcc\n"
        "v_cndmask_b32 %[w1], %[w1], %[w2], vcc\n"
        ".endr\n s_waitcnt lgkmcnt(0)"
        : [w0]"+v"(w0), [w1]"+v"(w1), [w2]"+v"(w2), [w3]"+v"(w3), [sm]"+v"(sm), [smk]"+v"(smk), [lz]"=&v"(lz), [p]"=&v"(p), [wb]"+v"(wb), [t0]"=&v"(u0), [t1]"=&v"(u1)
        : [sel]"s"(0x00010203u), [kp1]"v"(11u) : "vcc", "s20", "s21", "memory");
    a = w0 + w1 + w2 + w3 + sm;
BENCH_END

// no EXEC changes at all: the queue advance as selects, the LDS read unconditional (address selected)
BENCH_BEGIN(k_rice_step_v3)
    uint32_t w0 = a, w1 = b, w2 = c, w3 = d, sm = 5, smk = 3, lz, p, wb = threadIdx.x * 4, u0, u1;
    asm volatile(".rept " STR(REP) "\n"
        "v_alignbit_b32 %[p], %[w0], %[w1], %[sm]\n"
        "v_ffbh_u32 %[lz], %[p]\n"
        "v_and_b32 %[lz], 15, %[lz]\n"
        "v_cmp_lt_i32 vcc, %[smk], %[lz]\n"
        "v_perm_b32 %[t1], 0, %[w3], %[sel]\n"
        "v_cndmask_b32 %[w0], %[w0], %[w1], vcc\n"
        "v_cndmask_b32 %[w1], %[w1], %[w2], vcc\n"
        "v_cndmask_b32 %[w2], %[w2], %[t1], vcc\n"
        "v_addc_co_u32 %[wb], vcc, %[wb], %[wb], vcc\n"
        "v_and_b32 %[t0], 0x3fc, %[wb]\n"
        "ds_read_b32 %[w3], %[t0]\n"
        "v_sub_u32 %[sm], %[smk], %[lz]\n"
        "v_and_b32 %[sm], 31, %[sm]\n"
        "v_sub_u32 %[smk], %[sm], %[kp1]\n"
        ".endr\n s_waitcnt lgkmcnt(0)"
        : [w0]"+v"(w0), [w1]"+v"(w1), [w2]"+v"(w2), [w3]"+v"(w3), [sm]"+v"(sm), [smk]"+v"(smk), [lz]"=&v"(lz), [p]"=&v"(p), [wb]"+v"(wb), [t0]"=&v"(u0), [t1]"=&v"(u1)
        : [sel]"s"(0x00010203u), [kp1]"v"(11u) : "vcc", "memory");
    a = w0 + w1 + w2 + w3 + sm;
BENCH_END

// does a wave with few active lanes issue faster?  (EXEC = 16 / 32 lanes around a dependent chain)
BENCH_BEGIN(k_dep_add_exec16)
    asm volatile("s_mov_b64 exec, 0xffff\n .rept " STR(REP) "\n v_add_u32 %0, %0, %1\n .endr\n s_mov_b64 exec, -1" : "+v"(a) : "v"(b));
BENCH_END

BENCH_BEGIN(k_dep_add_exec32)
    asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 exec_hi, 0\n .rept " STR(REP) "\n v_add_u32 %0, %0, %1\n .endr\n s_mov_b64 exec, s[20:21]" : "+v"(a) : "v"(b) : "s20", "s21");
BENCH_END

BENCH_BEGIN(k_rice_step_v3_exec16)
    uint32_t w0 = a, w1 = b, w2 = c, w3 = d, sm = 5, smk = 3, lz, p, wb = threadIdx.x * 4, u0, u1;
    asm volatile("s_mov_b64 exec, 0xffff\n .rept " STR(REP) "\n"
        "v_alignbit_b32 %[p], %[w0], %[w1], %[sm]\n"
        "v_ffbh_u32 %[lz], %[p]\n"
        "v_and_b32 %[lz], 15, %[lz]\n"
        "v_cmp_lt_i32 vcc, %[smk], %[lz]\n"
        "v_perm_b32 %[t1], 0, %[w3], %[sel]\n"
        "v_cndmask_b32 %[w0], %[w0], %[w1], vcc\n"
        "v_cndmask_b32 %[w1], %[w1], %[w2], vcc\n"
        "v_cndmask_b32 %[w2], %[w2], %[t1], vcc\n"
        "v_addc_co_u32 %[wb], vcc, %[wb], %[wb], vcc\n"
        "v_and_b32 %[t0], 0x3fc, %[wb]\n"
        "ds_read_b32 %[w3], %[t0]\n"
        "v_sub_u32 %[sm], %[smk], %[lz]\n"
        "v_and_b32 %[sm], 31, %[sm]\n"
        "v_sub_u32 %[smk], %[sm], %[kp1]\n"
        ".endr\n s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, -1"
        : [w0]"+v"(w0), [w1]"+v"(w1), [w2]"+v"(w2), [w3]"+v"(w3), [sm]"+v"(sm), [smk]"+v"(smk), [lz]"=&v"(lz), [p]"=&v"(p), [wb]"+v"(wb), [t0]"=&v"(u0), [t1]"=&v"(u1)
        : [sel]"s"(0x00010203u), [kp1]"v"(11u) : "vcc", "memory");
    a = w0 + w1 + w2 + w3 + sm;
BENCH_END

// the delimiting step without the LDS read (how much of v3 is the read's latency?)
BENCH_BEGIN(k_rice_step_nolds)
    uint32_t w0 = a, w1 = b, w2 = c, w3 = d, sm = 5, smk = 3, lz, p, wb = threadIdx.x * 4, u0, u1;
    asm volatile(".rept " STR(REP) "\n"
        "v_alignbit_b32 %[p], %[w0], %[w1], %[sm]\n"
        "v_ffbh_u32 %[lz], %[p]\n"
        "v_and_b32 %[lz], 15, %[lz]\n"
        "v_cmp_lt_i32 vcc, %[smk], %[lz]\n"
        "v_perm_b32 %[t1], 0, %[w3], %[sel]\n"
        "v_cndmask_b32 %[w0], %[w0], %[w1], vcc\n"
        "v_cndmask_b32 %[w1], %[w1], %[w2], vcc\n"
        "v_cndmask_b32 %[w2], %[w2], %[t1], vcc\n"
        "v_addc_co_u32 %[wb], vcc, %[wb], %[wb], vcc\n"
        "v_and_b32 %[t0], 0x3fc, %[wb]\n"
        "v_xor_b32 %[w3], %[t0], %[w3]\n"
        "v_sub_u32 %[sm], %[smk], %[lz]\n"
        "v_and_b32 %[sm], 31, %[sm]\n"
        "v_sub_u32 %[smk], %[sm], %[kp1]\n"
        ".endr\n"
        : [w0]"+v"(w0), [w1]"+v"(w1), [w2]"+v"(w2), [w3]"+v"(w3), [sm]"+v"(sm), [smk]"+v"(smk), [lz]"=&v"(lz), [p]"=&v"(p), [wb]"+v"(wb), [t0]"=&v"(u0), [t1]"=&v"(u1)
        : [sel]"s"(0x00010203u), [kp1]"v"(11u) : "vcc", "memory");
    a = w0 + w1 + w2 + w3 + sm;
BENCH_END

// the decoder's delimiting step as shipped (flac_dec_fast.hip, FG_RC): two-word window, look-ahead read consumed by the last
// instruction of the next code
BENCH_BEGIN(k_rice_step_v5)
    uint32_t w0 = a, w1 = b, na = c, nb = d, t = 5, sm, lz, a4, pmin = 0xffffffffu, addr = threadIdx.x * 64;
    int32_t smk = 3;
    asm volatile(".rept " STR(REP) "\n"
        "v_alignbit_b32 v248, %[w0], %[w1], %[t]\n"
        "v_ffbh_u32 %[lz], v248\n"
        "v_cmp_lt_i32 vcc, %[smk], %[lz]\n"
        "v_cndmask_b32_e64 %[a4], 0, 4, vcc\n"
        "v_add_u32 %[addr], %[addr], %[a4]\n"
        "v_and_b32 %[addr], 0xffc, %[addr]\n"
        "ds_read_b32 %[nb], %[addr]\n"
        "v_sub_u32 %[t], %[smk], %[lz]\n"
        "v_and_b32 %[sm], 31, %[t]\n"
        "v_sub_u32 %[smk], %[sm], %[kp1]\n"
        "v_cndmask_b32 %[w0], %[w0], %[w1], vcc\n"
        "s_waitcnt lgkmcnt(1)\n"
        "v_cndmask_b32 %[w1], %[w1], %[na], vcc\n"
        "v_alignbit_b32 v249, %[w0], %[w1], %[t]\n"
        "v_ffbh_u32 %[lz], v249\n"
        "v_cmp_lt_i32 vcc, %[smk], %[lz]\n"
        "v_cndmask_b32_e64 %[a4], 0, 4, vcc\n"
        "v_add_u32 %[addr], %[addr], %[a4]\n"
        "v_and_b32 %[addr], 0xffc, %[addr]\n"
        "ds_read_b32 %[na], %[addr]\n"
        "v_min3_u32 %[pmin], %[pmin], v248, v249\n"
        "v_sub_u32 %[t], %[smk], %[lz]\n"
        "v_and_b32 %[sm], 31, %[t]\n"
        "v_sub_u32 %[smk], %[sm], %[kp1]\n"
        "v_cndmask_b32 %[w0], %[w0], %[w1], vcc\n"
        "s_waitcnt lgkmcnt(1)\n"
        "v_cndmask_b32 %[w1], %[w1], %[nb], vcc\n"
        ".endr\n s_waitcnt lgkmcnt(0)"
        : [w0]"+v"(w0), [w1]"+v"(w1), [na]"+v"(na), [nb]"+v"(nb), [t]"+v"(t), [smk]"+v"(smk), [pmin]"+v"(pmin), [addr]"+v"(addr), [sm]"=&v"(sm), [lz]"=&v"(lz), [a4]"=&v"(a4)
        : [kp1]"v"(11u) : "vcc", "v248", "v249", "memory");
    a = w0 + w1 + na + nb + pmin + t;
BENCH_END

BENCH_BEGIN(k_fmac64_dpp)
    asm volatile(".rept " STR(REP) "\n v_fmac_f64_dpp %0, %1, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n .endr" : "+v"(fa) : "v"(fb));
BENCH_END

BENCH_BEGIN(k_fma64_bcast_movs)
    uint32_t lo = a, hi = b, tl, th;
    // stands for: broadcast both halves, then a plain FMA on the pair (here fb twice, the moves are independent of it)
    asm volatile(".rept " STR(REP) "\n v_mov_b32_dpp %1, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fma_f64 %0, %5, %5, %0\n .endr" : "+v"(fa), "=&v"(tl), "=&v"(th) : "v"(lo), "v"(hi), "v"(fb));
    a += tl + th;
BENCH_END

BENCH_BEGIN(k_dpp_dep)
    asm volatile(".rept " STR(REP) "\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n .endr" : "+v"(a));
BENCH_END

BENCH_BEGIN(k_loop_branch)
    uint32_t s = REP;
    asm volatile("1:\n v_add_u32 %0, %0, %2\n s_sub_u32 %1, %1, 1\n s_cmp_lg_u32 %1, 0\n s_cbranch_scc1 1b" : "+v"(a), "+s"(s) : "v"(b) : "scc");
BENCH_END

// global pointer chase: buf[i] holds the byte offset of the next element
__global__ void __launch_bounds__(64) k_gchase(uint64_t *out, const uint32_t *chain, uint32_t steps, uint32_t *sink)
{
    uint32_t off = threadIdx.x == 0 ? 0 : 0;
    uint64_t t0 = clock64();
    for (uint32_t i = 0; i < steps; i++) off = *(const uint32_t *)((const char *)chain + off);
    uint64_t t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[threadIdx.x] = off;
}

typedef void (*kern_t)(uint64_t *, uint32_t *, uint32_t);

static double run(kern_t k, int nwg, uint64_t *d_out, uint32_t *d_buf)
{
    uint64_t h[4096];
    double best = 1e30;
    for (int it = 0; it < 3; it++) {
        hipLaunchKernelGGL(k, dim3(nwg), dim3(64), 0, 0, d_out, d_buf, 7u);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d_out, sizeof(uint64_t) * nwg, hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < nwg; i++) s += (double)h[i];
        s /= nwg;
        if (s < best) best = s;
    }
    return best;
}

int main()
{
    uint64_t *d_out; uint32_t *d_buf;
    CK(hipMalloc(&d_out, 4096 * 8)); CK(hipMalloc(&d_buf, 4096 * 4));
    struct { const char *name; kern_t k; int per; } tests[] = {
        {"dependent v_add_u32", k_dep_add, 1}, {"independent v_add_u32 (x2)", k_indep_add, 2},
        {"dependent alignbit+ffbh (x2)", k_dep_alignbit_ffbh, 2}, {"dependent v_mad_i32_i24", k_dep_mad24, 1},
        {"independent v_mad_i32_i24 (x2)", k_indep_mad24, 2}, {"dependent v_mad_i32_i16", k_dep_mad_i32_i16, 1}, {"dependent v_dot2_i32_i16", k_dep_dot2_i32_i16, 1}, {"dependent v_fma_f32", k_dep_fma_f32, 1}, {"recurrence sample: mul + 7 mad24 + shift + add (x10)", k_recur8, 10}, {"recurrence sample: 4 dot2 + shift + add (x6)", k_recur8_dot2, 6},
        {"dependent v_mul_lo_u32", k_dep_mul_lo, 1}, {"dependent v_mad_u64_u32", k_dep_mad_u64_u32, 1},
        {"dependent v_lshlrev_b64", k_dep_lshl64, 1}, {"dependent v_fma_f64", k_dep_fma64, 1}, {"independent v_fma_f64 (x2)", k_indep_fma64, 2},
        {"v_cmp + v_cndmask (x2)", k_cmp_cndmask, 2}, {"v_cmp + saveexec + v_add + s_or (x4)", k_cmp_saveexec, 4},
        {"v_cmp + saveexec + cbranch_execz + v_add + s_or (x5)", k_cmp_saveexec_branch, 5},
        {"dependent s_add_u32", k_salu_dep, 1}, {"v_add + s_add alternating (x2)", k_valu_salu_mix, 2},
        {"v_readlane + s_nop 3 + v_add (x3)", k_readlane_use, 3},
        {"ds_read_b32 dependent + wait", k_lds_read_dep, 1}, {"ds_write_b32 + wait", k_lds_write_wait, 1},
        {"ds_write_b32 back to back", k_lds_write_nowait, 1}, {"ds_write_b128 back to back", k_lds_write_b128_nowait, 1},
        {"ds_or_b32 back to back", k_lds_or_nowait, 1}, {"ds_or_b32 pair scattered (x2)", k_lds_or_pair_scattered, 2},
        {"ds_read_u16 back to back (stride 136B)", k_lds_read_u16_nowait, 1}, {"v_writelane + v_readlane (x2)", k_writelane_readlane, 2},
        {"v_mul_i32_i24 with SGPR operand", k_valu_sgpr_operand, 1},
        {"dependent v_fmac_f64_dpp row_newbcast", k_fmac64_dpp, 1}, {"2 x v_mov_b32_dpp + v_fma_f64 (x3)", k_fma64_bcast_movs, 3},
        {"rice step v1 (saveexec + skip branch)", k_rice_step_v1, 1}, {"rice step v2 (saveexec, no branch)", k_rice_step_v2, 1},
        {"rice step v3 (selects, unconditional LDS read)", k_rice_step_v3, 1},
        {"dependent v_add_u32, EXEC = 16 lanes", k_dep_add_exec16, 1}, {"dependent v_add_u32, EXEC = 32 lanes", k_dep_add_exec32, 1},
        {"rice step v3, EXEC = 16 lanes", k_rice_step_v3_exec16, 1}, {"rice step v3 without the LDS read", k_rice_step_nolds, 1},
        {"rice step v5 (shipped: 2 codes, + 2 wrap ands)", k_rice_step_v5, 2},
        {"dependent DPP v_add row_shr", k_dpp_dep, 1}, {"loop: v_add + s_sub + s_cmp + s_cbranch (x4)", k_loop_branch, 4},
    };
    for (int nwg : {1, 1280}) {
        const double base = run(k_empty, nwg, d_out, d_buf);
        printf("---- %d workgroups of one wave (empty: %.0f ticks)\n", nwg, base);
        for (auto &t : tests) {
            const double v = run(t.k, nwg, d_out, d_buf) - base;
            printf("%-56s %8.2f ticks per pattern, %6.2f per instruction\n", t.name, v / REP, v / REP / t.per);
        }
    }
    // global latency: stride through a buffer larger than L2 (chain of offsets)
    for (size_t bytes : {(size_t)1 << 16, (size_t)1 << 21, (size_t)1 << 26, (size_t)1 << 29}) {
        const size_t stride = 4096 + 128, n = bytes / stride;
        uint32_t *h = (uint32_t *)calloc(bytes / 4, 4);
        for (size_t i = 0; i < n; i++) h[i * stride / 4] = (uint32_t)(((i * 7919 + 1) % n) * stride);
        uint32_t *d_chain, *d_sink;
        CK(hipMalloc(&d_chain, bytes)); CK(hipMalloc(&d_sink, 256));
        CK(hipMemcpy(d_chain, h, bytes, hipMemcpyHostToDevice));
        uint64_t t = 0;
        for (int it = 0; it < 2; it++) {
            hipLaunchKernelGGL(k_gchase, dim3(1), dim3(64), 0, 0, d_out, d_chain, 2000u, d_sink);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost));
        }
        printf("global pointer chase over %8zu KiB: %.0f ticks per dependent load\n", bytes >> 10, (double)t / 2000);
        CK(hipFree(d_chain)); CK(hipFree(d_sink)); free(h);
    }
    // tick rate
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        uint64_t t = 0;
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_dep_fma64, dim3(1), dim3(64), 0, 0, d_out, d_buf, 7u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost));
        printf("one k_dep_fma64 launch: %.1f us wall (50 launches), %llu ticks inside\n", ms * 1000 / 50, (unsigned long long)t);
    }
    return 0;
}
