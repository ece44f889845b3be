// gfx950emu (test infrastructure, see emu.h): waves, workgroups, dispatches.
#pragma once
#include "emu.h"
#include <atomic>
#include <memory>

#define EMU_SHARED_HI 0x00010000u      // flat addresses: {aperture, offset}; both lie above any host pointer
#define EMU_PRIVATE_HI 0x00020000u

struct Dispatch;
struct Wave;
struct WG {
    u32 id[3] = {0, 0, 0};
    u32 cu = 0;                  // GFX950EMU_L1: which modelled vector L1 this workgroup reads through
    std::vector<u8> lds;
    u32 nwaves = 0, at_barrier = 0, done = 0;
    u64 barrier_gen = 0;
    std::vector<Wave *> members;
};
enum WaveState : u8 { W_RUN, W_BARRIER, W_DONE, W_FAULT };
struct Wave {
    Dispatch *d = nullptr;
    WG *wg = nullptr;
    u32 pc = 0;
    u64 exec = 0, vcc = 0;
    bool scc = false;
    u32 m0 = 0;
    u32 s[128];
    std::vector<u32> v, a;          // v[reg * 64 + lane]
    std::vector<u8> scratch;        // lane-major: scratch[lane * size + offset]
    WaveState state = W_RUN;
    u64 barrier_gen = 0;
    u64 sleep_until = 0;
    u8 gpr_idx_mode = 0;            // s_set_gpr_idx_on
    u32 gpr_idx = 0;
    u32 nv = 0, na = 0;
    int trace_lane = -1;            // GFX950EMU_WATCH
    u64 insts = 0;                  // instructions this wave has executed (KStats.max_wave_insts: a launch's longest serial path)
};
struct KStats {
    u64 wave_insts = 0, valu = 0, valu_lanes = 0, salu = 0, smem = 0, vmem = 0, lds = 0, mfma = 0, branch = 0, waves = 0;
    u64 global_load_bytes = 0, global_store_bytes = 0, launches = 0, max_wave_insts = 0;
};
struct Dispatch {
    CodeObject *co = nullptr;
    const KernelInfo *ki = nullptr;
    u32 grid[3] = {1, 1, 1}, block[3] = {1, 1, 1};      // grid in workgroups
    u32 lds_bytes = 0;
    std::vector<u8> kernarg;
    std::vector<u8> packet;                              // an hsa_kernel_dispatch_packet_t for kernels that read it
    u64 next_wg = 0, total_wgs = 0, finished_wgs = 0;
    std::vector<std::unique_ptr<WG>> wgs;                // resident
    std::vector<std::unique_ptr<Wave>> waves;            // resident
    bool failed = false;
    std::string error;
    KStats *stats = nullptr;
    std::vector<u64> *pc_hist = nullptr;                 // GFX950EMU_PROFILE: executions per instruction of the code object
};

// one instruction of one wave; false when the wave cannot go on right now (barrier, done, fault)
bool emu_step(Wave &w);
// (hipshim.cpp) is [p, p + n) memory the device may touch?
bool emu_mem_ok(u64 p, u64 n);
extern std::atomic<u64> g_emu_clock;      // wall clock, 100 MHz ticks (advanced by the scheduler)
void emu_fault(Wave &w, const char *fmt, ...);
// GFX950EMU_L1=<n>: a model of the per-CU vector L1 as the weakest thing the hardware may do -- a line a plain load brought in is served
// from there, whatever other CUs store, until a kernel starts or a wave of that CU executes buffer_inv sc1; sc1 / nt loads and atomics go
// past it; the CU's own stores update it.  n CUs, workgroups dealt to them round-robin (few CUs: more sharing, more staleness).
extern u32 g_emu_l1_cus;
bool emu_l1_read(u32 cu, u64 addr, void *dst, u32 n);        // false: not cacheable here (the line leaves an allocation): read memory
void emu_l1_store(u32 cu, u64 addr, const void *src, u32 n);
void emu_l1_drop_line(u32 cu, u64 addr, u32 n);
void emu_l1_invalidate(u32 cu);
void emu_l1_invalidate_all();
bool emu_l1_sc1_bypasses();
