// What a LONE wave pays per instruction in dependent chains of the kinds the restore kernel's recurrence is made of (tuning aid;
// not part of the product).  build: hipcc --offload-arch=gfx950 -O2 -o chainlat chainlat.hip
// 256 workgroups of one wave (one per CU), each runs REPS x a block of one kind; time by wall_clock64 (100 MHz) inside the wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define R8(S) S S S S S S S S
template <int KIND>
__global__ void __launch_bounds__(64) k_chain(uint32_t *sink, unsigned long long *ticks, int reps)
{
    uint32_t a = threadIdx.x + 1, b = threadIdx.x * 2654435761u | 1u, c = threadIdx.x + 12345u, d = 7, e = 9, f = 11, g = 13, sh = 3, sel = 0x05040100u;
    uint32_t x0 = 1, x1 = 2, x2 = 3, x3 = 4, x4 = 5, x5 = 6, x6 = 7, x7 = 8, t = 0, p = 0, pn = 0, n = 0;
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
    for (int r = 0; r < reps; r++) {
        if (KIND == 0) {          // 56 dependent v_mad_i32_i24
            asm volatile(R8("v_mad_i32_i24 %0, %1, %2, %0\n" "v_mad_i32_i24 %0, %1, %2, %0\n" "v_mad_i32_i24 %0, %1, %2, %0\n" "v_mad_i32_i24 %0, %1, %2, %0\n"
                            "v_mad_i32_i24 %0, %1, %2, %0\n" "v_mad_i32_i24 %0, %1, %2, %0\n" "v_mad_i32_i24 %0, %1, %2, %0\n") : "+v"(a) : "v"(b), "v"(c));
        }
        else if (KIND == 1) {     // 56 dependent v_dot2_i32_i16
            asm volatile(R8("v_dot2_i32_i16 %0, %1, %2, %0\n" "v_dot2_i32_i16 %0, %1, %2, %0\n" "v_dot2_i32_i16 %0, %1, %2, %0\n" "v_dot2_i32_i16 %0, %1, %2, %0\n"
                            "v_dot2_i32_i16 %0, %1, %2, %0\n" "v_dot2_i32_i16 %0, %1, %2, %0\n" "v_dot2_i32_i16 %0, %1, %2, %0\n") : "+v"(a) : "v"(b), "v"(c));
        }
        else if (KIND == 2) {     // 56 v_dot2_i32_i16 on seven accumulators (independent neighbours)
            asm volatile(R8("v_dot2_i32_i16 %0, %7, %8, %0\n" "v_dot2_i32_i16 %1, %7, %8, %1\n" "v_dot2_i32_i16 %2, %7, %8, %2\n" "v_dot2_i32_i16 %3, %7, %8, %3\n"
                            "v_dot2_i32_i16 %4, %7, %8, %4\n" "v_dot2_i32_i16 %5, %7, %8, %5\n" "v_dot2_i32_i16 %6, %7, %8, %6\n")
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(x0), "+v"(x1) : "v"(b), "v"(c));
        }
        else if (KIND == 3) {     // the recurrence as it is (round 5): 4 dependent dot2, shift, SDWA add, perm -- eight samples
            asm volatile(R8("v_dot2_i32_i16 %0, %1, %2, 0\n" "v_dot2_i32_i16 %0, %3, %4, %0\n" "v_dot2_i32_i16 %0, %5, %6, %0\n" "v_dot2_i32_i16 %0, %7, %8, %0\n"
                            "v_ashrrev_i32 %0, %9, %0\n"
                            "v_add_u32_sdwa %10, sext(%11), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
                            "v_perm_b32 %1, %1, %10, %12\n")
                         : "+v"(t), "+v"(x7), "+v"(b), "+v"(x5), "+v"(c), "+v"(x3), "+v"(d), "+v"(x1), "+v"(e) : "v"(sh), "v"(n), "v"(f), "v"(sel));
            asm volatile("" : "+v"(n));
        }
        else if (KIND == 4) {     // the same sums, the three older pairs gathered for the NEXT sample between the steps of this one
            asm volatile(R8("v_dot2_i32_i16 %0, %1, %2, %13\n" "v_dot2_i32_i16 %14, %3, %4, 0\n" "v_ashrrev_i32 %0, %9, %0\n" "v_dot2_i32_i16 %14, %5, %6, %14\n"
                            "v_add_u32_sdwa %10, sext(%11), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
                            "v_dot2_i32_i16 %13, %7, %8, %14\n"
                            "v_perm_b32 %1, %1, %10, %12\n")
                         : "+v"(t), "+v"(x7), "+v"(b), "+v"(x5), "+v"(c), "+v"(x3), "+v"(d), "+v"(x1), "+v"(e), "+v"(sh), "+v"(n), "+v"(f), "+v"(sel), "+v"(p), "+v"(pn));
        }
        else if (KIND == 5) {     // 56 dependent v_add_u32
            asm volatile(R8("v_add_u32 %0, %1, %0\n" "v_add_u32 %0, %1, %0\n" "v_add_u32 %0, %1, %0\n" "v_add_u32 %0, %1, %0\n"
                            "v_add_u32 %0, %1, %0\n" "v_add_u32 %0, %1, %0\n" "v_add_u32 %0, %1, %0\n") : "+v"(a) : "v"(b));
        }
        else if (KIND == 6) {     // the newest sample through v_mad_i32_i24 (the perm off the chain): 8 instructions a sample
            asm volatile(R8("v_mad_i32_i24 %0, %10, %2, %13\n" "v_dot2_i32_i16 %14, %3, %4, 0\n" "v_ashrrev_i32 %0, %9, %0\n" "v_dot2_i32_i16 %14, %5, %6, %14\n"
                            "v_add_u32_sdwa %10, sext(%11), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
                            "v_dot2_i32_i16 %14, %7, %8, %14\n"
                            "v_perm_b32 %1, %1, %10, %12\n"
                            "v_dot2_i32_i16 %13, %1, %8, %14\n")
                         : "+v"(t), "+v"(x7), "+v"(b), "+v"(x5), "+v"(c), "+v"(x3), "+v"(d), "+v"(x1), "+v"(e), "+v"(sh), "+v"(n), "+v"(f), "+v"(sel), "+v"(p), "+v"(pn));
        }
        else if (KIND == 7) {     // 7 dependent: dot2, ashr, sdwa add, perm + 3 dependent dot2 -- as 3 but on a chain of mads (round 4's form, 10 a sample)
            asm volatile(R8("v_mul_i32_i24 %0, %1, %2\n" "v_mad_i32_i24 %0, %3, %4, %0\n" "v_mad_i32_i24 %0, %5, %6, %0\n" "v_mad_i32_i24 %0, %7, %8, %0\n"
                            "v_mad_i32_i24 %0, %3, %2, %0\n" "v_mad_i32_i24 %0, %5, %4, %0\n" "v_mad_i32_i24 %0, %7, %6, %0\n" "v_mad_i32_i24 %0, %1, %8, %0\n"
                            "v_ashrrev_i32 %0, %9, %0\n"
                            "v_add_u32_sdwa %1, sext(%11), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n")
                         : "+v"(t), "+v"(x7), "+v"(b), "+v"(x5), "+v"(c), "+v"(x3), "+v"(d), "+v"(x1), "+v"(e) : "v"(sh), "v"(n), "v"(f), "v"(sel));
        }
        else if (KIND == 8 || KIND == 9) {     // as 3, with the LDS traffic of the kernel: two 16-byte writes and one 16-byte read a group of eight
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            extern __shared__ u4 lds4[];
            // (rows of 256 bytes, the 16-byte slot XORed with the row as in the kernel: no bank conflicts)
            u4 *wp = lds4 + threadIdx.x * 16 + ((((uint32_t)r & 7) * 2) ^ (threadIdx.x & 14));
            u4 w; w.x = x7; w.y = x5; w.z = x3; w.w = x1;
            wp[0] = w; wp[1] = w;
            if (KIND == 9) { wp[1024] = w; wp[1025] = w; }
            const u4 rd = lds4[2048 + threadIdx.x * 8 + (((uint32_t)r & 7) ^ (threadIdx.x & 7))];
            asm volatile(R8("v_dot2_i32_i16 %0, %1, %2, 0\n" "v_dot2_i32_i16 %0, %3, %4, %0\n" "v_dot2_i32_i16 %0, %5, %6, %0\n" "v_dot2_i32_i16 %0, %7, %8, %0\n"
                            "v_ashrrev_i32 %0, %9, %0\n"
                            "v_add_u32_sdwa %10, sext(%11), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
                            "v_perm_b32 %1, %1, %10, %12\n")
                         : "+v"(t), "+v"(x7), "+v"(b), "+v"(x5), "+v"(c), "+v"(x3), "+v"(d), "+v"(x1), "+v"(e) : "v"(sh), "v"(n), "v"(f), "v"(sel));
            f += rd.x;
        }
    }
    const unsigned long long c1 = clock64();
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = c1 - c0; }
    sink[blockIdx.x * 64 + threadIdx.x] = a + t + x7 + x5 + x3 + x1 + d + e + f + g + x0 + n + p + pn + b + c;
}

template <int KIND>
static void run(const char *name, int per_block, uint32_t *sink, unsigned long long *ticks, int reps, int blocks)
{
    hipLaunchKernelGGL(k_chain<KIND>, dim3(blocks), dim3(64), 65536, 0, sink, ticks, reps);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_chain<KIND>, dim3(blocks), dim3(64), 65536, 0, sink, ticks, reps);
    CK(hipDeviceSynchronize());
    unsigned long long h[2 * 1024];
    CK(hipMemcpy(h, ticks, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
    double tk = 0, ck = 0;
    for (int i = 0; i < blocks; i++) { tk += (double)h[2 * i]; ck += (double)h[2 * i + 1]; }
    tk /= blocks; ck /= blocks;
    const double ns = tk * 10.0 / ((double)reps * per_block);
    printf("%-64s %6.2f ns an instruction (%5.2f cycles at 2.4 GHz; s_memtime %5.2f counts)  [%d waves]\n", name, ns, ns * 2.4, ck / ((double)reps * per_block), blocks);
}

int main(int argc, char **argv)
{
    const int reps = 20000;
    uint32_t *sink; unsigned long long *ticks;
    CK(hipMalloc(&sink, 1024 * 64 * 4)); CK(hipMalloc(&ticks, 1024 * 2 * 8));
    for (int blocks = 220; blocks <= 220; blocks += 1) {
        run<5>("dependent v_add_u32", 56, sink, ticks, reps, blocks);
        run<0>("dependent v_mad_i32_i24", 56, sink, ticks, reps, blocks);
        run<1>("dependent v_dot2_i32_i16", 56, sink, ticks, reps, blocks);
        run<2>("v_dot2_i32_i16, seven accumulators in turn", 56, sink, ticks, reps, blocks);
        run<7>("recurrence of round 4 (mul + 7 mad + shift + add), 10 a sample", 80, sink, ticks, reps, blocks);
        run<3>("recurrence of round 5 (4 dot2 + shift + add + perm), 7 a sample", 56, sink, ticks, reps, blocks);
        run<4>("  the older pairs gathered a sample ahead, between the steps", 56, sink, ticks, reps, blocks);
        run<8>("  round 5's with its LDS traffic (2 x ds_write_b128, 1 x ds_read_b128 per 56)", 56, sink, ticks, reps, blocks);
        run<9>("  ... with 4 x ds_write_b128", 56, sink, ticks, reps, blocks);
        run<6>("  newest sample by v_mad_i32_i24, perm off the chain, 8 a sample", 64, sink, ticks, reps, blocks);
    }
    return 0;
}
