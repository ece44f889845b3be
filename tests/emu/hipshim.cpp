// gfx950emu (test infrastructure, see emu.h): the HIP runtime entry points libflacgpu.so imports, served by the interpreter.
// Built as tests/emu/libamdhip64.so.7 and loaded IN FRONT OF the product library by the emulator tests only.
//
// Device memory is host memory (hipMalloc = aligned_alloc, filled with 0xCD so that a kernel that relies on fresh memory being zero
// shows).  Streams are in-order queues; ONE device thread runs the kernels at the head of all streams interleaved, wave by wave, so
// that a kernel waiting for a word another stream's kernel raises makes progress, and the host thread runs beside it as it does
// beside a GPU (it polls pinned memory the kernels write).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include "exec.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

std::atomic<u64> g_emu_clock{1000};

namespace {
struct Range { u64 lo, hi; };
struct EmuEvent { u64 record_seq = 0, done_seq = 0, stamp = 0; };
struct QOp {
    enum T { KERNEL, MEMCPY, MEMSET, RECORD, WAIT } t;
    Dispatch *d = nullptr;
    void *dst = nullptr; const void *src = nullptr; size_t n = 0; int value = 0;
    std::vector<u8> staged;
    EmuEvent *ev = nullptr; u64 seq = 0;
};
struct EmuStream { std::deque<QOp> q; Dispatch *running = nullptr; };
struct FuncRef { CodeObject *co; std::string name; };

struct Global {
    std::mutex mu;
    std::condition_variable cv_host, cv_dev;
    std::vector<EmuStream *> streams;
    EmuStream null_stream;
    std::map<u64, u64> ranges;                 // lo -> hi
    std::atomic<u64> ranges_version{1};
    std::vector<std::unique_ptr<CodeObject>> cos;
    std::map<const void *, FuncRef> funcs;
    std::map<std::string, KStats> stats;
    std::thread dev;
    bool dev_started = false, quit = false;
    int sticky_error = 0;
    std::string sticky_msg;
    u64 quantum = 128;
    size_t max_waves = 4096;
    bool poison = true, trace = false;
    u64 sched_seed = 0;
    std::string profile_kernel; std::map<std::string, std::pair<CodeObject *, std::vector<u64>>> profiles;     // GFX950EMU_PROFILE=substring
    std::string watch_kernel; u64 watch_wg = 0; u32 watch_wave = 0; int watch_lane = 0;     // GFX950EMU_WATCH=substring:wg:wave:lane
};
Global &G() { static Global *g = new Global; return *g; }
thread_local int t_last_error = 0;

// the device thread's own copy of the allocation table
std::vector<Range> d_ranges;
u64 d_ranges_version = 0;
void refresh_ranges()
{
    Global &g = G();
    const u64 v = g.ranges_version.load(std::memory_order_acquire);
    if (v == d_ranges_version) return;
    std::lock_guard<std::mutex> lk(g.mu);
    d_ranges.clear();
    for (auto &r : g.ranges) d_ranges.push_back({r.first, r.second});
    d_ranges_version = g.ranges_version.load();
}
void add_range(const void *p, size_t n) { Global &g = G(); g.ranges[(u64)(uintptr_t)p] = (u64)(uintptr_t)p + n; g.ranges_version.fetch_add(1, std::memory_order_release); }
void del_range(const void *p) { Global &g = G(); g.ranges.erase((u64)(uintptr_t)p); g.ranges_version.fetch_add(1, std::memory_order_release); }
}  // namespace

// ---------------------------------------------------------------------------------------------------------------- the L1 model
u32 g_emu_l1_cus = 0;
bool emu_l1_sc1_bypasses() { extern bool emu_l1_ignore_sc1_flag(); return !emu_l1_ignore_sc1_flag(); }
namespace {
struct L1Line { u8 b[64]; };
std::vector<std::unordered_map<u64, L1Line>> g_l1;
bool g_l1_ignore_sc1 = false;          // GFX950EMU_L1_IGNORE_SC1=1: sc1 loads are served from the L1 like plain ones (what code WITHOUT agent-scope loads would see)
bool g_l1_ignore_inv = false;          // GFX950EMU_L1_IGNORE_INV=1: buffer_inv does nothing (what a kernel WITHOUT its acquire fence would see)
u64 g_l1_wg_counter = 0;
}
bool emu_mem_ok(u64 p, u64 n);
bool emu_l1_read(u32 cu, u64 addr, void *dst, u32 n)
{
    u8 *out = (u8 *)dst;
    u64 a = addr;
    u32 left = n;
    while (left) {
        const u64 line = a & ~63ull;
        const u32 off = (u32)(a - line), take = std::min<u32>(left, 64 - off);
        auto &m = g_l1[cu];
        auto it = m.find(line);
        if (it == m.end()) {
            if (!emu_mem_ok(line, 64)) return false;
            L1Line ln; memcpy(ln.b, (const void *)(uintptr_t)line, 64);
            it = m.emplace(line, ln).first;
        }
        memcpy(out, it->second.b + off, take);
        out += take; a += take; left -= take;
    }
    return true;
}
void emu_l1_store(u32 cu, u64 addr, const void *src, u32 n)
{
    const u8 *in = (const u8 *)src;
    u64 a = addr;
    u32 left = n;
    while (left) {
        const u64 line = a & ~63ull;
        const u32 off = (u32)(a - line), take = std::min<u32>(left, 64 - off);
        auto it = g_l1[cu].find(line);
        if (it != g_l1[cu].end()) memcpy(it->second.b + off, in, take);
        in += take; a += take; left -= take;
    }
}
void emu_l1_drop_line(u32 cu, u64 addr, u32 n) { for (u64 line = addr & ~63ull; line < addr + n; line += 64) g_l1[cu].erase(line); }
void emu_l1_invalidate(u32 cu) { if (!g_l1_ignore_inv) g_l1[cu].clear(); }
void emu_l1_invalidate_all() { for (auto &m : g_l1) m.clear(); }
bool emu_l1_ignore_sc1_flag() { return g_l1_ignore_sc1; }

bool emu_mem_ok(u64 p, u64 n)
{
    static thread_local size_t last = 0;
    if (last < d_ranges.size() && p >= d_ranges[last].lo && p + n <= d_ranges[last].hi) return true;
    // binary search
    size_t a = 0, b = d_ranges.size();
    while (a < b) { const size_t m = (a + b) / 2; if (d_ranges[m].lo <= p) a = m + 1; else b = m; }
    if (a == 0) { refresh_ranges(); return false; }
    if (p + n <= d_ranges[a - 1].hi) { last = a - 1; return true; }
    return false;
}

// ---------------------------------------------------------------------------------------------------------------- dispatch
static void init_wave(Dispatch &d, WG &wg, Wave &w, u32 wave_idx, u32 threads_in_wg)
{
    const KernelInfo &ki = *d.ki;
    w.d = &d; w.wg = &wg; w.pc = ki.first_inst;
    w.nv = std::max<u32>(ki.vgprs, 8); w.na = ki.agprs;
    w.v.assign((size_t)(w.nv + 8) * 64, 0); w.a.assign((size_t)(w.na + 2) * 64, 0);
    if (ki.scratch) w.scratch.assign((size_t)ki.scratch * 64, 0xCD);
    memset(w.s, 0, sizeof w.s);
    // lanes of this wave
    const u32 first = wave_idx * 64, nl = std::min<u32>(64, threads_in_wg - first);
    w.exec = nl == 64 ? ~0ull : ((1ull << nl) - 1);
    for (u32 l = 0; l < nl; l++) {
        const u32 t = first + l;
        const u32 x = t % d.block[0], y = (t / d.block[0]) % d.block[1], z = t / (d.block[0] * d.block[1]);
        w.v[l] = x | (y << 10) | (z << 20);
    }
    // user SGPRs in the order of the descriptor's enable bits, then the workgroup ids
    u32 s = 0;
    const u32 props = ki.props;
    if (props & 1) s += 4;                                         // private segment buffer (not used with architected scratch)
    if (props & 2) { const u64 pa = (u64)(uintptr_t)d.packet.data(); w.s[s] = (u32)pa; w.s[s + 1] = (u32)(pa >> 32); s += 2; }         // dispatch packet
    if (props & 4) s += 2;                                         // queue
    if (props & 8) { const u64 ka = (u64)(uintptr_t)d.kernarg.data(); w.s[s] = (u32)ka; w.s[s + 1] = (u32)(ka >> 32); s += 2; }
    if (props & 16) s += 2;
    if (props & 32) s += 2;
    if (props & 64) s += 1;
    const u32 user = (ki.rsrc2 >> 1) & 0x1F;
    if (user > s) s = user;
    if (ki.rsrc2 >> 7 & 1) w.s[s++] = wg.id[0];
    if (ki.rsrc2 >> 8 & 1) w.s[s++] = wg.id[1];
    if (ki.rsrc2 >> 9 & 1) w.s[s++] = wg.id[2];
    w.state = W_RUN;
}

static u64 g_sched_state = 0;
static u64 sched_rand()       // xorshift64*
{
    u64 x = g_sched_state;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    g_sched_state = x;
    return x * 2685821657736338717ull;
}

static void activate(Dispatch &d)
{
    Global &g = G();
    const u32 threads = d.block[0] * d.block[1] * d.block[2], nw = (threads + 63) / 64;
    while (d.next_wg < d.total_wgs && d.waves.size() + nw <= g.max_waves) {
        std::unique_ptr<WG> wg(new WG);
        const u64 id = d.next_wg++;
        wg->id[0] = (u32)(id % d.grid[0]); wg->id[1] = (u32)((id / d.grid[0]) % d.grid[1]); wg->id[2] = (u32)(id / ((u64)d.grid[0] * d.grid[1]));
        wg->lds.assign(d.lds_bytes, g.poison ? 0xCD : 0);
        if (g_emu_l1_cus) wg->cu = (u32)(g_l1_wg_counter++ % g_emu_l1_cus);
        wg->nwaves = nw;
        for (u32 k = 0; k < nw; k++) {
            std::unique_ptr<Wave> w(new Wave);
            init_wave(d, *wg, *w, k, threads);
            if (!g.watch_kernel.empty() && d.ki->name.find(g.watch_kernel) != std::string::npos && id == g.watch_wg && k == g.watch_wave) w->trace_lane = g.watch_lane;
            wg->members.push_back(w.get());
            d.waves.push_back(std::move(w));
            if (d.stats) d.stats->waves++;
        }
        d.wgs.push_back(std::move(wg));
    }
}

static void release_barrier(WG &wg)
{
    if (wg.at_barrier == 0 || wg.at_barrier < wg.nwaves - wg.done) return;
    for (Wave *m : wg.members) if (m->state == W_BARRIER) m->state = W_RUN;
    wg.at_barrier = 0;
}

// one turn of every resident wave; returns the wave-instructions executed
static u64 run_slice(Dispatch &d)
{
    Global &g = G();
    activate(d);
    u64 n = 0;
    bool any_done = false;
    // GFX950EMU_SCHED=<seed>: an adversarial scheduler -- the waves of a slice in random order, each for a random number of
    // instructions (1 .. 2 quantum), some skipped altogether: orderings between workgroups (look-back words, tickets, the join) that
    // the fixed round-robin never produces
    std::vector<u32> order;
    if (g.sched_seed) {
        order.resize(d.waves.size());
        for (u32 i = 0; i < order.size(); i++) order[i] = i;
        for (size_t i = order.size(); i > 1; i--) { const size_t j = (size_t)(sched_rand() % i); std::swap(order[i - 1], order[j]); }
    }
    for (size_t ii = 0; ii < d.waves.size(); ii++) {
        const size_t i = g.sched_seed ? order[ii] : ii;
        Wave &w = *d.waves[i];
        if (w.state != W_RUN) continue;
        u64 q = 0, lim = g.quantum;
        if (g.sched_seed) { const u64 r = sched_rand(); if ((r & 7) == 0) continue; lim = 1 + (r >> 8) % (2 * g.quantum); }
        while (q < lim && emu_step(w)) q++;
        n += q + 1;
        if (w.state == W_BARRIER) release_barrier(*w.wg);
        else if (w.state == W_DONE || w.state == W_FAULT) { w.wg->done++; any_done = true; release_barrier(*w.wg); }
        if (d.failed) break;
    }
    if (d.failed) { d.waves.clear(); d.wgs.clear(); d.finished_wgs = d.total_wgs; d.next_wg = d.total_wgs; return n; }
    if (any_done) {
        // retire whole workgroups
        std::vector<std::unique_ptr<Wave>> keep;
        for (auto &w : d.waves) if (w->wg->done < w->wg->nwaves) keep.push_back(std::move(w));
        d.waves.swap(keep);
        std::vector<std::unique_ptr<WG>> keepg;
        for (auto &wg : d.wgs) { if (wg->done >= wg->nwaves) d.finished_wgs++; else keepg.push_back(std::move(wg)); }
        d.wgs.swap(keepg);
    }
    return n;
}

static void device_main()
{
    Global &g = G();
    std::vector<std::pair<EmuStream *, Dispatch *>> active;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(g.mu);
            for (;;) {
                active.clear();
                bool progressed = false;
                std::vector<EmuStream *> all = g.streams;
                all.push_back(&g.null_stream);
                for (EmuStream *s : all) {
                    while (!s->running && !s->q.empty()) {
                        QOp &op = s->q.front();
                        if (op.t == QOp::KERNEL) { s->running = op.d; s->q.pop_front(); progressed = true; if (g_emu_l1_cus) emu_l1_invalidate_all(); break; }
                        if (op.t == QOp::MEMCPY) { if (op.n) memmove(op.dst, op.staged.empty() ? op.src : op.staged.data(), op.n); }
                        else if (op.t == QOp::MEMSET) memset(op.dst, op.value, op.n);
                        else if (op.t == QOp::RECORD) { op.ev->done_seq = std::max(op.ev->done_seq, op.seq); op.ev->stamp = g_emu_clock.load(); }
                        else if (op.t == QOp::WAIT) { if (op.ev->done_seq < op.seq) break; }
                        s->q.pop_front();
                        progressed = true;
                    }
                    if (s->running) active.push_back({s, s->running});
                }
                if (progressed) g.cv_host.notify_all();
                if (!active.empty() || g.quit) break;
                if (!progressed) g.cv_dev.wait(lk);
            }
            if (g.quit) return;
        }
        refresh_ranges();
        u64 n = 0;
        if (g.sched_seed && active.size() > 1) for (size_t i = active.size(); i > 1; i--) std::swap(active[i - 1], active[(size_t)(sched_rand() % i)]);
        for (auto &a : active) n += run_slice(*a.second);
        g_emu_clock.fetch_add(std::max<u64>(1, n / 64), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(g.mu);
            bool fin = false;
            for (auto &a : active) {
                Dispatch *d = a.second;
                if (d->finished_wgs >= d->total_wgs) {
                    if (d->failed && !g.sticky_error) { g.sticky_error = hipErrorIllegalAddress; g.sticky_msg = d->error; }
                    if (g.trace) fprintf(stderr, "gfx950emu: done   %s\n", d->ki->name.c_str());
                    a.first->running = nullptr;
                    del_range(d->kernarg.data()); del_range(d->packet.data());
                    delete d;
                    fin = true;
                }
            }
            if (fin) g.cv_host.notify_all();
        }
    }
}

static void ensure_device()
{
    Global &g = G();
    if (g.dev_started) return;
    g.dev_started = true;
    if (const char *q = getenv("GFX950EMU_QUANTUM")) g.quantum = (u64)atoll(q);
    if (const char *q = getenv("GFX950EMU_MAX_WAVES")) g.max_waves = (size_t)atoll(q);
    if (const char *q = getenv("GFX950EMU_POISON")) g.poison = atoi(q) != 0;
    g.trace = getenv("GFX950EMU_TRACE") != nullptr;
    if (const char *sd = getenv("GFX950EMU_SCHED")) { g.sched_seed = (u64)atoll(sd); g_sched_state = g.sched_seed * 0x9E3779B97F4A7C15ull + 1; }
    if (const char *pk = getenv("GFX950EMU_PROFILE")) g.profile_kernel = pk;
    if (const char *l1 = getenv("GFX950EMU_L1")) { g_emu_l1_cus = (u32)atoi(l1); g_l1.resize(g_emu_l1_cus); }
    g_l1_ignore_inv = getenv("GFX950EMU_L1_IGNORE_INV") != nullptr;
    g_l1_ignore_sc1 = getenv("GFX950EMU_L1_IGNORE_SC1") != nullptr;
    if (const char *wv = getenv("GFX950EMU_WATCH")) {
        std::string t = wv; unsigned long long a = 0; unsigned b = 0; int c = 0;
        const size_t p1 = t.find(':');
        g.watch_kernel = t.substr(0, p1);
        if (p1 != std::string::npos) sscanf(t.c_str() + p1 + 1, "%llu:%u:%d", &a, &b, &c);
        g.watch_wg = a; g.watch_wave = b; g.watch_lane = c;
    }
    g.dev = std::thread(device_main);
    g.dev.detach();
}
static EmuStream *S(hipStream_t s) { return s ? (EmuStream *)s : &G().null_stream; }
static void enqueue(hipStream_t s, QOp &&op)
{
    Global &g = G();
    std::lock_guard<std::mutex> lk(g.mu);
    ensure_device();
    S(s)->q.push_back(std::move(op));
    g.cv_dev.notify_all();
}
static int wait_stream(EmuStream *s)
{
    Global &g = G();
    std::unique_lock<std::mutex> lk(g.mu);
    g.cv_host.wait(lk, [&] { return s->q.empty() && !s->running; });
    return g.sticky_error;
}
static int wait_all()
{
    Global &g = G();
    std::unique_lock<std::mutex> lk(g.mu);
    g.cv_host.wait(lk, [&] {
        if (!g.null_stream.q.empty() || g.null_stream.running) return false;
        for (EmuStream *s : g.streams) if (!s->q.empty() || s->running) return false;
        return true;
    });
    return g.sticky_error;
}
static hipError_t E(int e) { if (e) t_last_error = e; return (hipError_t)e; }

// ---------------------------------------------------------------------------------------------------------------- statistics
extern "C" void gfx950emu_reset_stats(void) { Global &g = G(); std::lock_guard<std::mutex> lk(g.mu); for (auto &s : g.stats) s.second = KStats(); }
// JSON: per kernel name, what its launches executed since the last reset
extern "C" const char *gfx950emu_stats_json(void)
{
    static std::string out;
    Global &g = G();
    std::lock_guard<std::mutex> lk(g.mu);
    out = "{";
    bool first = true;
    for (auto &s : g.stats) {
        const KStats &k = s.second;
        if (!k.launches) continue;
        char b[1024];
        snprintf(b, sizeof b, "%s\"%s\": {\"launches\": %llu, \"waves\": %llu, \"wave_insts\": %llu, \"valu\": %llu, \"valu_lanes\": %llu, \"salu\": %llu, \"smem\": %llu, "
                 "\"vmem\": %llu, \"lds\": %llu, \"mfma\": %llu, \"global_load_bytes\": %llu, \"global_store_bytes\": %llu, \"max_wave_insts\": %llu}", first ? "" : ", ", s.first.c_str(),
                 (unsigned long long)k.launches, (unsigned long long)k.waves, (unsigned long long)k.wave_insts, (unsigned long long)k.valu, (unsigned long long)k.valu_lanes,
                 (unsigned long long)k.salu, (unsigned long long)k.smem, (unsigned long long)k.vmem, (unsigned long long)k.lds, (unsigned long long)k.mfma,
                 (unsigned long long)k.global_load_bytes, (unsigned long long)k.global_store_bytes, (unsigned long long)k.max_wave_insts);
        out += b; first = false;
    }
    out += "}";
    return out.c_str();
}
// GFX950EMU_PROFILE=<substring of a kernel name>: executions per instruction, written as "count line-of-the-disassembly mnemonic" rows
// (the disassembly: $GFX950EMU_CACHE/<hash>.s), one file per kernel
extern "C" int gfx950emu_write_profiles(const char *dir)
{
    Global &g = G();
    std::lock_guard<std::mutex> lk(g.mu);
    int n = 0;
    for (auto &p : g.profiles) {
        char path[512]; snprintf(path, sizeof path, "%s/profile_%d.txt", dir, n++);
        FILE *f = fopen(path, "w");
        if (!f) continue;
        fprintf(f, "# %s\n", p.first.c_str());
        const auto &h = p.second.second;
        for (size_t i = 0; i < h.size(); i++) if (h[i]) fprintf(f, "%llu %u %s\n", (unsigned long long)h[i], p.second.first->insts[i].line, op_name(p.second.first->insts[i].op));
        fclose(f);
    }
    return n;
}
extern "C" const char *gfx950emu_last_fault(void) { static std::string s; Global &g = G(); std::lock_guard<std::mutex> lk(g.mu); s = g.sticky_msg; return s.c_str(); }
extern "C" void gfx950emu_clear_fault(void) { Global &g = G(); std::lock_guard<std::mutex> lk(g.mu); g.sticky_error = 0; g.sticky_msg.clear(); }

// ---------------------------------------------------------------------------------------------------------------- the HIP entry points
extern "C" {

struct FatWrapper { u32 magic, version; const void *binary; void *unused; };
void **__hipRegisterFatBinary(const void *data)
{
    Global &g = G();
    const FatWrapper *fw = (const FatWrapper *)data;
    const u8 *b = (const u8 *)fw->binary;
    std::vector<CodeObject *> *mods = new std::vector<CodeObject *>;
    if (memcmp(b, "__CLANG_OFFLOAD_BUNDLE__", 24) != 0) { fprintf(stderr, "gfx950emu: fat binary is not a clang offload bundle\n"); return (void **)mods; }
    u64 cnt; memcpy(&cnt, b + 24, 8);
    const u8 *q = b + 32;
    for (u64 i = 0; i < cnt; i++) {
        u64 off, size, tl; memcpy(&off, q, 8); memcpy(&size, q + 8, 8); memcpy(&tl, q + 16, 8); q += 24;
        const std::string triple((const char *)q, tl); q += tl;
        if (triple.find("gfx950") == std::string::npos || size == 0) continue;
        std::unique_ptr<CodeObject> co(new CodeObject);
        if (!load_code_object(*co, b + off, size)) continue;
        std::lock_guard<std::mutex> lk(g.mu);
        add_range(co->image.data(), co->image.size());
        mods->push_back(co.get());
        g.cos.push_back(std::move(co));
    }
    return (void **)mods;
}
void __hipRegisterFunction(void **modules, const void *hostFunction, char *deviceFunction, const char *deviceName, int, void *, void *, void *, void *, int *)
{
    Global &g = G();
    std::vector<CodeObject *> *mods = (std::vector<CodeObject *> *)modules;
    (void)deviceFunction;
    for (CodeObject *co : *mods) if (co->kernels.count(deviceName)) { std::lock_guard<std::mutex> lk(g.mu); g.funcs[hostFunction] = FuncRef{co, deviceName}; return; }
    fprintf(stderr, "gfx950emu: kernel %s not found in its code object\n", deviceName);
}
void __hipRegisterVar(void **, void *, char *, const char *name, int, size_t, int, int) { fprintf(stderr, "gfx950emu: device variable %s not supported\n", name); }
void __hipUnregisterFatBinary(void **) {}

struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
static thread_local std::vector<CallCfg> t_cfg;
hipError_t __hipPushCallConfiguration(dim3 gridDim, dim3 blockDim, size_t sharedMem, hipStream_t stream) { t_cfg.push_back({gridDim, blockDim, sharedMem, stream}); return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *gridDim, dim3 *blockDim, size_t *sharedMem, hipStream_t *stream)
{
    if (t_cfg.empty()) return E(hipErrorInvalidValue);
    const CallCfg c = t_cfg.back(); t_cfg.pop_back();
    *gridDim = c.grid; *blockDim = c.block; *sharedMem = c.shmem; *stream = c.stream;
    return hipSuccess;
}

hipError_t hipLaunchKernel(const void *hostFunction, dim3 gridDim, dim3 blockDim, void **args, size_t sharedMem, hipStream_t stream)
{
    Global &g = G();
    FuncRef fr;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        auto it = g.funcs.find(hostFunction);
        if (it == g.funcs.end()) return E(hipErrorInvalidDeviceFunction);
        fr = it->second;
        if (!fr.co->parsed && !parse_text(*fr.co)) return E(hipErrorInvalidImage);
    }
    const KernelInfo &ki = fr.co->kernels[fr.name];
    if ((u64)gridDim.x * gridDim.y * gridDim.z == 0 || blockDim.x * blockDim.y * blockDim.z == 0 || blockDim.x * blockDim.y * blockDim.z > 1024) return E(hipErrorInvalidConfiguration);
    if (ki.lds_static + sharedMem > 160 * 1024) return E(hipErrorInvalidValue);
    Dispatch *d = new Dispatch;
    d->co = fr.co; d->ki = &ki;
    d->grid[0] = gridDim.x; d->grid[1] = gridDim.y; d->grid[2] = gridDim.z;
    d->block[0] = blockDim.x; d->block[1] = blockDim.y; d->block[2] = blockDim.z;
    d->total_wgs = (u64)gridDim.x * gridDim.y * gridDim.z;
    d->lds_bytes = (u32)(ki.lds_static + sharedMem);
    d->kernarg.assign(ki.kernarg_size + 64, 0);
    u32 explicit_i = 0;
    for (const KernArg &a : ki.args) {
        u8 *p = d->kernarg.data() + a.offset;
        auto put = [&](u64 v, u32 n) { memcpy(p, &v, n); };
        if (a.kind.compare(0, 7, "hidden_") != 0) { memcpy(p, args[explicit_i++], a.size); continue; }
        if (a.kind == "hidden_block_count_x") put(gridDim.x, 4); else if (a.kind == "hidden_block_count_y") put(gridDim.y, 4); else if (a.kind == "hidden_block_count_z") put(gridDim.z, 4);
        else if (a.kind == "hidden_group_size_x") put(blockDim.x, 2); else if (a.kind == "hidden_group_size_y") put(blockDim.y, 2); else if (a.kind == "hidden_group_size_z") put(blockDim.z, 2);
        else if (a.kind == "hidden_grid_dims") put((gridDim.z > 1 || blockDim.z > 1) ? 3 : (gridDim.y > 1 || blockDim.y > 1) ? 2 : 1, 2);
        else if (a.kind == "hidden_dynamic_lds_size") put(sharedMem, 4);
        else if (a.kind == "hidden_shared_base") put(EMU_SHARED_HI, 4); else if (a.kind == "hidden_private_base") put(EMU_PRIVATE_HI, 4);
        // (remainders, global offsets, printf / hostcall buffers, queue pointers: zero)
    }
    {
        // hsa_kernel_dispatch_packet_t: workgroup sizes (u16 at 4, 6, 8), grid sizes in work-items (u32 at 12, 16, 20), segment sizes, kernarg address
        d->packet.assign(64, 0);
        u8 *pk = d->packet.data();
        const u16 wsz[3] = {(u16)blockDim.x, (u16)blockDim.y, (u16)blockDim.z};
        const u32 gsz[3] = {gridDim.x * blockDim.x, gridDim.y * blockDim.y, gridDim.z * blockDim.z};
        memcpy(pk + 4, wsz, 6); memcpy(pk + 12, gsz, 12);
        memcpy(pk + 24, &ki.scratch, 4); memcpy(pk + 28, &d->lds_bytes, 4);
        const u64 ka = (u64)(uintptr_t)d->kernarg.data(); memcpy(pk + 40, &ka, 8);
    }
    {
        std::lock_guard<std::mutex> lk(g.mu);
        add_range(d->packet.data(), d->packet.size());
        add_range(d->kernarg.data(), d->kernarg.size());      // (until the dispatch retires)
        KStats &st = g.stats[fr.name];
        st.launches++;
        d->stats = &st;
        if (!g.profile_kernel.empty() && fr.name.find(g.profile_kernel) != std::string::npos) {
            auto &pr = g.profiles[fr.name];
            pr.first = fr.co;
            if (pr.second.size() != fr.co->insts.size()) pr.second.assign(fr.co->insts.size(), 0);
            d->pc_hist = &pr.second;
        }
        if (g.trace) fprintf(stderr, "gfx950emu: launch %s grid %u block %u lds %u\n", fr.name.c_str(), gridDim.x, blockDim.x, d->lds_bytes);
    }
    QOp op; op.t = QOp::KERNEL; op.d = d;
    enqueue(stream, std::move(op));
    return hipSuccess;
}

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : E(hipErrorInvalidDevice); }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *pi, hipDeviceAttribute_t attr, int)
{
    switch (attr) {
    case hipDeviceAttributeWallClockRate: *pi = 100000; break;          // kHz
    case hipDeviceAttributeMultiprocessorCount: *pi = 256; break;
    case hipDeviceAttributeMaxSharedMemoryPerBlock: *pi = 160 * 1024; break;
    case hipDeviceAttributeWarpSize: *pi = 64; break;
    case hipDeviceAttributeClockRate: *pi = 2400000; break;
    default: *pi = 0; break;
    }
    return hipSuccess;
}
hipError_t hipGetLastError(void) { const int e = t_last_error; t_last_error = 0; return (hipError_t)e; }
const char *hipGetErrorString(hipError_t e)
{
    if (e == hipSuccess) return "no error";
    static thread_local char b[600];
    snprintf(b, sizeof b, "gfx950emu error %d%s%s", (int)e, G().sticky_msg.empty() ? "" : ": ", G().sticky_msg.c_str());
    return b;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

hipError_t hipMalloc(void **p, size_t n)
{
    Global &g = G();
    const size_t sz = (n + 255) & ~(size_t)255;
    void *m = aligned_alloc(256, sz ? sz : 256);
    if (!m) return E(hipErrorOutOfMemory);
    if (g.poison) memset(m, 0xCD, sz ? sz : 256);
    std::lock_guard<std::mutex> lk(g.mu);
    add_range(m, sz ? sz : 256);
    *p = m;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    wait_all();
    { std::lock_guard<std::mutex> lk(G().mu); del_range(p); }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { return hipFree(p); }
// (memory a test allocated itself -- a numpy array handed over as "device" memory -- can be made known with this)
void gfx950emu_register(const void *p, size_t n) { std::lock_guard<std::mutex> lk(G().mu); add_range(p, n); }
void gfx950emu_unregister(const void *p) { wait_all(); std::lock_guard<std::mutex> lk(G().mu); del_range(p); }

hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind)
{
    const int e = wait_all();
    if (n) memmove(dst, src, n);
    return E(e);
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t stream)
{
    QOp op; op.t = QOp::MEMCPY; op.dst = dst; op.src = src; op.n = n;
    // a source outside every registered range is pageable host memory: staged now, as the runtime does
    bool known;
    { Global &g = G(); std::lock_guard<std::mutex> lk(g.mu); auto it = g.ranges.upper_bound((u64)(uintptr_t)src); known = it != g.ranges.begin() && (--it)->second >= (u64)(uintptr_t)src + n; }
    if (!known && n) op.staged.assign((const u8 *)src, (const u8 *)src + n);
    bool dst_known;
    { Global &g = G(); std::lock_guard<std::mutex> lk(g.mu); auto it = g.ranges.upper_bound((u64)(uintptr_t)dst); dst_known = it != g.ranges.begin() && (--it)->second >= (u64)(uintptr_t)dst + n; }
    enqueue(stream, std::move(op));
    if (!dst_known) return E(wait_stream(S(stream)));      // (to pageable host memory: done when the call returns)
    return hipSuccess;
}
hipError_t hipMemset(void *dst, int v, size_t n) { const int e = wait_all(); memset(dst, v, n); return E(e); }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t stream) { QOp op; op.t = QOp::MEMSET; op.dst = dst; op.value = v; op.n = n; enqueue(stream, std::move(op)); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int)
{
    Global &g = G();
    EmuStream *es = new EmuStream;
    std::lock_guard<std::mutex> lk(g.mu);
    g.streams.push_back(es);
    *s = (hipStream_t)es;
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s) return hipSuccess;
    wait_stream(S(s));
    Global &g = G();
    std::lock_guard<std::mutex> lk(g.mu);
    for (size_t i = 0; i < g.streams.size(); i++) if (g.streams[i] == (EmuStream *)s) { g.streams.erase(g.streams.begin() + i); break; }
    delete (EmuStream *)s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { return E(s ? wait_stream(S(s)) : wait_all()); }
hipError_t hipDeviceSynchronize(void) { return E(wait_all()); }
hipError_t hipStreamQuery(hipStream_t s)
{
    Global &g = G();
    std::lock_guard<std::mutex> lk(g.mu);
    EmuStream *es = S(s);
    return (es->q.empty() && !es->running) ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t) new EmuEvent; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { wait_all(); delete (EmuEvent *)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    EmuEvent *ev = (EmuEvent *)e;
    QOp op; op.t = QOp::RECORD; op.ev = ev;
    { std::lock_guard<std::mutex> lk(G().mu); op.seq = ++ev->record_seq; }
    enqueue(s, std::move(op));
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int)
{
    EmuEvent *ev = (EmuEvent *)e;
    QOp op; op.t = QOp::WAIT; op.ev = ev;
    { std::lock_guard<std::mutex> lk(G().mu); op.seq = ev->record_seq; }
    if (op.seq == 0) return hipSuccess;           // never recorded: nothing to wait for
    enqueue(s, std::move(op));
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    EmuEvent *ev = (EmuEvent *)e;
    Global &g = G();
    std::unique_lock<std::mutex> lk(g.mu);
    g.cv_host.wait(lk, [&] { return ev->done_seq >= ev->record_seq; });
    return E(g.sticky_error);
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    EmuEvent *ea = (EmuEvent *)a, *eb = (EmuEvent *)b;
    std::lock_guard<std::mutex> lk(G().mu);
    if (ea->done_seq < ea->record_seq || eb->done_seq < eb->record_seq) return E(hipErrorNotReady);
    *ms = (float)((double)(eb->stamp - ea->stamp) / 1e5);
    return hipSuccess;
}
const char *hipGetErrorName(hipError_t e) { return hipGetErrorString(e); }

}  // extern "C"
