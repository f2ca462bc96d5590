// emei_device.h — device-side helpers shared by the env kernels (gfx950 / wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/emei_hip.h"
#include "emei_math.h"

namespace emei {

constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU
constexpr int kWave = 64;

// Device event counters for tools/newton_stats.py and tools/pend_stats.py (a variant build with -DEMEI_NEWTON_STATS only;
// nothing in a normal build): one array per translation unit, read back through emei_debug_stats_<tu>() of body_tu.hip /
// pendulum_tu.hip.  Newton solve (cheetah_model.h, hopper_model.h):
//   0 evaluations (lanes)   1 evaluations with rows (lanes)   2 Newton passes (lanes)   3 Newton passes (waves)
//   4 contact-row blocks executed (waves)   5 contact-row blocks (lanes)   6 limit-row blocks (waves)   7 evaluations (waves)
//   8 + k: lane evaluations that took k passes (k >= 13 in the last bin)
//   22 / 23 evaluations on the constraint-space path (lanes / waves)   24 / 25 on the primal loop (lanes / waves)
// staged pendulum kernels (pendulum_kernels.h, pendulum_envs.h):
//   16 env-steps (waves)   17 ... in which some lane resets   18 ... in which spares are redrawn
//   19 substeps (waves)   20 ... that run the slider-limit block   21 lanes beyond the rail
// -DEMEI_CLOCK_PROBE (tools/clock_probe.py, round 4; a variant build, nothing else changes): every wave of a rollout kernel
// reads the shader clock (s_memtime) and the constant 100 MHz counter (s_memrealtime) when it starts and when it ends;
// slots 28 / 29 / 30 = sum of shader cycles, sum of 100 MHz ticks, waves.  Their ratio is the clock the kernel really ran at.
#if defined(EMEI_CLOCK_PROBE) && !defined(EMEI_NEWTON_STATS)
static __device__ unsigned long long g_debug_stats[32];
#endif
#ifdef EMEI_CLOCK_PROBE
#ifndef EMEI_CLOCK_BIN_TICKS
#define EMEI_CLOCK_BIN_TICKS 5000  // 50 us
#endif
struct ClockProbe {
    unsigned long long c0, r0;
    __device__ __forceinline__ void begin() { c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void end() const {
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) {
            atomicAdd(&g_debug_stats[28], c1 - c0), atomicAdd(&g_debug_stats[29], r1 - r0), atomicAdd(&g_debug_stats[30], 1ull);
            // round 5 (tools/tail_probe.py): the span of the launch(es) in the same 100 MHz ticks — latest end in slot 27, earliest
            // begin as the maximum of its complement in slot 31 (both start from the cleared 0) — so that utilisation is a ratio
            // of two readings of ONE counter; slot 26: ticks spent waiting for a predecessor work item (body_kernels.h:WorkQueue)
            atomicMax(&g_debug_stats[27], r1), atomicMax(&g_debug_stats[31], ~r0);
            atomicMax(&g_debug_stats[23], r1 - r0);  // the longest single lifetime
#ifndef EMEI_NEWTON_STATS
            // histogram of lifetimes in EMEI_CLOCK_BIN_TICKS bins (slots 0-15; tools/pend_span.py): are the waves of a SIMD served fairly?
#ifdef EMEI_CLOCK_HIST_WAVEID  // ... or of the waves' slot ids on their SIMD (HW_REG_HW_ID bits [3:0])
            atomicAdd(&g_debug_stats[__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 15u], 1ull);
#elif defined(EMEI_CLOCK_HIST_CORR)  // slots 0-5 belong to body_kernels.h's item-to-item correlation
            (void)0;
#elif defined(EMEI_CLOCK_HIST_XCC)  // ... or lifetimes per XCD: slots 0-7 ticks, 8-15 waves (HW_REG_XCC_ID = 20, bits [3:0])
            {
                const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;
                atomicAdd(&g_debug_stats[xcc], r1 - r0), atomicAdd(&g_debug_stats[8 + xcc], 1ull);
            }
#else
            atomicAdd(&g_debug_stats[(r1 - r0) / EMEI_CLOCK_BIN_TICKS < 15 ? (r1 - r0) / EMEI_CLOCK_BIN_TICKS : 15], 1ull);
#endif
#endif
        }
    }
    // a persistent worker leaves (no ticket left): slot 21 workers, slot 22 the sum of their exit ticks — the mean exit against the
    // last end (slot 27) is the drain of the launch
    __device__ __forceinline__ static void worker_exit() {
        const unsigned long long r = __builtin_amdgcn_s_memrealtime();
        if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) atomicAdd(&g_debug_stats[21], 1ull), atomicAdd(&g_debug_stats[22], r);
    }
    // persistent workers (body_kernels.h:WorkQueue): how many of them got at least one item (slot 24), and when the last of them
    // began its first (slot 25, the same ticks): workers that are not resident from the start of the launch show up here
    __device__ __forceinline__ void first_item() const {
        if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) atomicAdd(&g_debug_stats[24], 1ull), atomicMax(&g_debug_stats[25], r0);
    }
    __device__ __forceinline__ void waited(unsigned long long ticks) const {
        if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) atomicAdd(&g_debug_stats[26], ticks);
    }
};
#define EMEI_CLOCK_BEGIN() emei::ClockProbe clock_probe_; clock_probe_.begin()
#define EMEI_CLOCK_END() clock_probe_.end()
#define EMEI_CLOCK_WAITED(t) clock_probe_.waited(t)
#define EMEI_CLOCK_FIRST_ITEM() clock_probe_.first_item()
#define EMEI_CLOCK_WORKER_EXIT() emei::ClockProbe::worker_exit()
#else
#define EMEI_CLOCK_BEGIN() ((void)0)
#define EMEI_CLOCK_END() ((void)0)
#define EMEI_CLOCK_WAITED(t) ((void)0)
#define EMEI_CLOCK_FIRST_ITEM() ((void)0)
#define EMEI_CLOCK_WORKER_EXIT() ((void)0)
#endif
#ifdef EMEI_NEWTON_STATS
static __device__ unsigned long long g_debug_stats[32];
// Branch-free on purpose: one atomic per ACTIVE lane with an addend of 1 (LANE) or of 1 for the first active lane and 0 for the
// others (WAVE).  Build the variant with `-mllvm -amdgpu-atomic-optimizer-strategy=None` (tools/build_variant.sh does it for
// -DEMEI_NEWTON_STATS): LLVM's atomic optimizer otherwise rewrites each atomic as "one lane adds the wave's sum" behind an
// s_and_saveexec, and round 2's statistics build of the Hopper RK4 kernel was MISCOMPILED at exactly such a join (spill
// stores in front of the EXEC restore: every lane but one per wave went non-finite; profiles/r03_hopper_rk4_stats_diag.txt).
#define EMEI_STAT_LANE(i) atomicAdd(&emei::g_debug_stats[i], 1ull)
#define EMEI_STAT_WAVE(i) \
    atomicAdd(&emei::g_debug_stats[i], (unsigned long long)((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1))
#else
#define EMEI_STAT_LANE(i) ((void)0)
#define EMEI_STAT_WAVE(i) ((void)0)
#endif

// Region markers.  Nothing in a normal build.  -DEMEI_ISA_MARKS (tools/isa_regions.py): a comment line in the ISA, for static
// instruction counts per region.  -DEMEI_CYCLE_PROFILE (tools/cycle_profile.py, round 4): each mark charges the shader-clock
// cycles since the wave's previous mark to the region that ENDS here — one atomic by the wave's first active lane, the running
// (timestamp, region) pair in LDS (one-wave blocks: the body rollout kernels of the one-wave-per-SIMD bodies) — so that a run
// reports where a wave's time goes, divergent regions and waits included.  COARSE by construction: a mark costs ~500 cycles of
// its own (a scalar clock read and three dependent LDS operations by one lane) and hipcc moves pure arithmetic across it (the
// factorisation that source code places behind `nw_smooth0` is hoisted in front of it), so only regions of several thousand
// cycles mean anything.  Region ids: enum emei_region below.
enum emei_region {
    EMEI_R_entry = 0, EMEI_R_nw_trig, EMEI_R_nw_forces, EMEI_R_nw_rows, EMEI_R_nw_direct, EMEI_R_nw_smooth0, EMEI_R_dual_fill,
    EMEI_R_dual_gram, EMEI_R_dual_loop, EMEI_R_dual_final, EMEI_R_nw_pass_base, EMEI_R_nw_limits, EMEI_R_nw_contacts, EMEI_R_nw_conv,
    EMEI_R_nw_step, EMEI_R_nw_final, EMEI_R_nw_euler, EMEI_R_nw_out, EMEI_R_step_io, EMEI_R_step_reset, EMEI_R_hp_pairs, EMEI_R_hp_verify, EMEI_R_tri_gram, EMEI_R_tri_loop, EMEI_R_tri_final, EMEI_R_count
};
#if defined(EMEI_ISA_MARKS)
#define EMEI_MARK(name) asm volatile("; EMEI_MARK " #name)
#elif defined(EMEI_CYCLE_PROFILE)
static __device__ unsigned long long g_cycle_stats[32];
struct CycleProfile {  // LDS of a one-wave block: per-region sums of this launch, the running (timestamp, region) pair
    unsigned long long acc[32], last;
    int prev;
};
__device__ __forceinline__ volatile CycleProfile* emei_cycle_lds() {
    __shared__ CycleProfile p;
    return &p;
}
__device__ __forceinline__ void emei_cycle_begin() {
    volatile CycleProfile* p = emei_cycle_lds();
    if (threadIdx.x < 32) p->acc[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) p->last = __builtin_readcyclecounter(), p->prev = EMEI_R_entry;
    __syncthreads();
}
__device__ __forceinline__ void emei_cycle_mark(int region) {
    volatile CycleProfile* p = emei_cycle_lds();
    const unsigned long long now = __builtin_readcyclecounter();
    if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) {  // LDS only: ~100 cycles per mark
        p->acc[p->prev] += now - p->last;
        p->last = now, p->prev = region;
    }
}
__device__ __forceinline__ void emei_cycle_end() {  // every thread of the block reaches this
    volatile CycleProfile* p = emei_cycle_lds();
    emei_cycle_mark(EMEI_R_entry);
    __syncthreads();
    if (threadIdx.x < EMEI_R_count) atomicAdd(&g_cycle_stats[threadIdx.x], p->acc[threadIdx.x]);
}
#define EMEI_MARK(name) emei::emei_cycle_mark(emei::EMEI_R_##name)
#define EMEI_PROFILE_BEGIN() emei::emei_cycle_begin()
#define EMEI_PROFILE_END() emei::emei_cycle_end()
#else
#define EMEI_MARK(name) ((void)0)
#endif
#ifndef EMEI_PROFILE_BEGIN
#define EMEI_PROFILE_BEGIN() ((void)0)
#define EMEI_PROFILE_END() ((void)0)
#endif

// ---------------------------------------------------------------------------------------------
// Lanes of ONE wave exchanging data through their private LDS slice need no s_barrier (a wave's LDS
// operations execute in order), but the COMPILER must not move a lane's read above another lane's
// write: per-thread alias analysis sees "this thread wrote bytes [16l,16l+16) and reads [4l,4l+4)" as
// disjoint and may hoist the read (it did, in the float32 build of the body kernel).  A
// wavefront-scope release/acquire pair around a wave barrier pins the order; it costs at most an
// s_waitcnt lgkmcnt(0).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 16-byte output stores of the staged rollout kernels with the NON-TEMPORAL hint (global_store_dwordx4 ... nt).  The rollout's
// outputs are written once and read by nobody on this GPU during the launch; without the hint the reward / done flushes in particular
// (256 B and 64 B pieces of a row per wave) sit in the L2 as partly written lines.  Same-box A/B (tools/ab.sh, kernel ms per pass):
// CartPoleSwingUp 65 536 envs 0.2614 -> 0.2343 (observation stores alone: 0.2585), 131 072 envs 0.5718 -> 0.4618, CartPoleBalancing
// 0.1477 -> 0.1305, InvertedPendulum 0.5328 -> 0.5175 (profiles/EXPERIMENTS.md).  Round 1 had tried the hint on the observation
// stores alone and seen nothing.
#ifndef EMEI_NT_STORES
#define EMEI_NT_STORES 1  // 0: a variant build with plain output stores, for A/B runs
#endif
__device__ __forceinline__ void store16_stream(float4* p, const float4& v) {
#if EMEI_NT_STORES
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const v4f_ t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (v4f_*)p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void store16_stream(uint4* p, const uint4& v) {
#if EMEI_NT_STORES
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    const v4u_ t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (v4u_*)p);
#else
    *p = v;
#endif
}

// ... and the body kernels' per-step outputs (linear float4 observation stores, 4 B rewards, 1 B done flags per lane) where the body says so
// (Body::kStreamOutputs: the light bodies — InvertedDoublePendulum 0.6404 -> 0.6056 ms; the cheetah and the Hopper, which are nowhere near
// the memory system's limits, do not gain: +0.7 % / 0)
template <bool STREAM, typename T>
__device__ __forceinline__ void store_body_out(T* p, const T& v) {
    if constexpr (STREAM && EMEI_NT_STORES != 0) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool STREAM>
__device__ __forceinline__ void store_body_out(float4* p, const float4& v) {
    if constexpr (STREAM) store16_stream(p, v);
    else *p = v;
}

// ---------------------------------------------------------------------------------------------
// trigonometry in the precision of the env.  Fast path: emei_math.h (straight-line, ~35 instructions);
// the device library's Payne-Hanek sincos only repairs the (practically unreachable) |x| > 1e6 case,
// AFTER the straight-line code, so that the hot basic block is not split.
__device__ __forceinline__ void sincos_fast_r(double x, double& s, double& c) { fast_sincos(x, s, c); }
__device__ __forceinline__ void sincos_fast_r(float x, float& s, float& c) { fast_sincosf(x, s, c); }
// repair of a fast result for out-of-range arguments; call it at the END of a straight-line block
// The cold path is entered through a WAVE-UNIFORM test (a scalar branch on the ballot), and masks its lanes inside.  The
// plain `if (cold)` form compiles to `s_and_saveexec; s_cbranch_execnz COLD; JOIN: ...; s_or_b64 exec`, and hipcc may
// put register spill stores at the top of JOIN, in front of the EXEC restore, where they execute with EXEC = 0 in every
// normal step (the cheetah RK4 miscompile; tests/test_isa_guards.py audits the shipped kernels for it).
// Round 3: inside the cold region the repair is computed by EVERY lane and SELECTED per lane — no `if (cold)`, hence no
// s_and_saveexec / EXEC-restore join in the region at all (the same allocator bug struck a -DEMEI_NEWTON_STATS build at
// such a join, tools/isa_scan.py shape (b)); the library's large-argument reduction loops run under their own masks.
__device__ __forceinline__ void sincos_repair_r(double x, double& s, double& c) {
    const bool cold = fabs(x) > kFastTrigLimitF64;
    if (__builtin_expect(__ballot(cold) != 0ull, 0)) {
        double s2, c2;
        ::sincos(x, &s2, &c2);
        s = cold ? s2 : s, c = cold ? c2 : c;
    }
}
__device__ __forceinline__ void sincos_repair_r(float x, float& s, float& c) {
    const bool cold = fabsf(x) > kFastTrigLimitF32;
    if (__builtin_expect(__ballot(cold) != 0ull, 0)) {
        float s2, c2;
        ::sincosf(x, &s2, &c2);
        s = cold ? s2 : s, c = cold ? c2 : c;
    }
}
template <typename T>
__device__ __forceinline__ void sincos_r(T x, T& s, T& c) {
    sincos_fast_r(x, s, c);
    sincos_repair_r(x, s, c);
}

// ---------------------------------------------------------------------------------------------
// Arguments beyond the fast path's range: exact reduction modulo 2 pi (Payne-Hanek on the hardware's 2/pi segments,
// v_trig_preop_f64), so that the SAME table path serves every finite angle.  The device library's sincos() used to
// repair such arguments after the fact; inlined, its ~160 instructions and 8 polynomial constants (16 VGPRs, hoisted
// out of the rollout loop) sat in every kernel for a branch that no physical trajectory takes.  This reduction needs
// five constants and returns an angle in [-4, 4] with an absolute error <= ~1e-15 (NaN for a non-finite argument).
__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}
__device__ __forceinline__ double trig_reduce_large(double x) {
    const double ax = __builtin_fabs(x);
    // three consecutive 53-bit pieces of 2/pi positioned for the exponent of x: the bits in front of p2 only contribute
    // whole multiples of 4 quadrants to x * 2/pi.  Above 2^945 the hardware returns the pieces scaled by 2^128.
    const double p2 = __builtin_amdgcn_trig_preop(ax, 0), p1 = __builtin_amdgcn_trig_preop(ax, 1),
                 p0 = __builtin_amdgcn_trig_preop(ax, 2);
    const double xs = ax >= 0x1.0p+945 ? __builtin_ldexp(ax, -128) : ax;
    double f2h = p2 * xs;
    const double f2l = __builtin_fma(p2, xs, -f2h);
    const double f1h = p1 * xs, f1l = __builtin_fma(p1, xs, -f1h);
    const double f0h = p0 * xs;
    // y = x * 2/pi modulo 4 as an unevaluated sum: top term modulo 4 (exact), then two-sums
    const double q = __builtin_ldexp(f2h, -2);
    f2h = __builtin_ldexp(__builtin_amdgcn_fract(q), 2);
    double s1, e1, s2, e2;
    two_sum(f1h, f2l, s1, e1);
    two_sum(f2h, s1, s2, e2);
    const double low = ((f1l + f0h) + e1) + e2;
    const double k = __builtin_rint(s2);
    const double f = (s2 - k) + low;                       // |f| <= 1/2: the fraction of a quadrant
    const double k4 = __builtin_fma(-4.0, __builtin_rint(k * 0.25), k);  // quadrant, centred: -2 .. 2
    const double t = k4 + f;
    const double r = __builtin_fma(t, 6.123233995736766e-17, t * 1.5707963267948966);
    const double res = x < 0.0 ? -r : r;
    return __builtin_isfinite(x) ? res : __builtin_nan("");
}
// the argument the table path sees: x itself in every practical case
__device__ __forceinline__ double trig_arg(double x) {
    const bool cold = !(__builtin_fabs(x) <= kFastTrigLimitF64);
    double r = x;
    if (__builtin_expect(__ballot(cold) != 0ull, 0)) {  // wave-uniform entry: see sincos_repair_r
        const double red = trig_reduce_large(x);       // straight-line arithmetic: every lane computes it, cold lanes take it
        r = cold ? red : x;
    }
    return r;
}

// Trig context of the cart/pole kernels: float64 uses the 256-entry {sin,cos} table every kernel
// stages in LDS (emei_math.h: fast_sincos_tab); float32 keeps the polynomial kernels.
struct TrigCtx {
    const SinCosEntry* tab;  // LDS
    // two polynomial coefficients kept in VGPRs for the whole kernel (sincos_begin_ctx): with both coefficients of an
    // inner Horner step as literals hipcc emits v_mov_b64 + the two-address v_fmac_f64 per evaluation
    double c3, c4;  // -1/6, 1/24
    uint32_t lds_base;  // byte address of `tab` in LDS, wave-uniform (an SGPR)
    // per-block LDS scratch of the body kernels (Body::kScratchPerLane elements of Body::real per lane, lane-interleaved:
    // element e of this lane at scratch[e * scratch_stride + threadIdx.x]); null where a kernel provides none
    void* scratch = nullptr;
    int scratch_stride = kBlock;  // threads per block of the kernel that owns the scratch (a compile-time value after inlining)
    // handle counter of Newton solves that ended at the iteration cap without meeting their stopping rule (emei_get_solver_cap_hits);
    // null in the stateless kernels
    unsigned long long* cap_hits = nullptr;
};
// The unit-step Newton iteration of the multi-constraint bodies has no line search (cheetah_model.h:accel_newton): it is the
// active-set iteration of a strictly convex piecewise-quadratic cost and ends in a handful of passes, but nothing PROVES it
// cannot cycle.  A lane that reaches the cap is therefore COUNTED: the parity tests assert the counter stays 0.
// Wave-uniform entry, the lanes add 0 / 1: no divergent branch in the hot kernel.
__device__ __forceinline__ void report_cap_hit(const TrigCtx& t, bool hit) {
    if (__builtin_expect(__ballot(hit) != 0ull, 0)) {
        if (t.cap_hits) atomicAdd(t.cap_hits, hit ? 1ull : 0ull);
    }
}
__device__ __forceinline__ void trig_ctx_init(TrigCtx& t, const SinCosEntry* tab) {
    t.tab = tab;
    t.c3 = -1.0 / 6, t.c4 = 1.0 / 24;
    asm volatile("" : "+v"(t.c3), "+v"(t.c4));  // opaque: not re-materialised at every use
    uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)tab);
    asm volatile("" : "+s"(base));  // opaque: as a literal the base costs the index a third instruction (v_or_b32)
    t.lds_base = base;
}
// a * B + c with the constant B in scalar registers and c in vector registers: the three-address form
__device__ __forceinline__ double fma_vsv(double a, double b_uniform, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
    return d;
}
// a * b + c, all in vector registers, three-address (c survives): for an addend that is still needed afterwards
__device__ __forceinline__ double fma_vvv(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_vvv(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float fma_vsv(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ void sincos_fast_ctx(const TrigCtx& t, double x, double& s, double& c) {
    fast_sincos_tab(x, t.tab, s, c);  // |x| <= kFastTrigLimitF64 (callers reduce first: sincos_ctx, sincos_begin_ctx)
}
__device__ __forceinline__ void sincos_fast_ctx(const TrigCtx&, float x, float& s, float& c) { fast_sincosf(x, s, c); }
// Two-phase form for the substep: begin as soon as the new angle exists, end after the dynamics, so
// that the table read's LDS latency (~100 cycles with bank conflicts) hides under ~25 float64
// operations.  hipcc sinks an ordinary LDS load down to its first use and is free to move pure
// arithmetic across a sched_barrier at the IR level, so the read and its wait are written as asm
// and pinned by data dependencies: `pin(a, b)` makes (a, b) depend on the issued read (work that
// consumes them cannot be hoisted above it), `end(after0, after1)` waits only once those values exist.
// An asm LDS read is invisible to hipcc's lgkmcnt bookkeeping, which is safe here: LDS operations of a
// wave return in order, so an uncounted extra read can only make a compiler-placed wait conservative.
typedef double emei_d2 __attribute__((ext_vector_type(2)));
struct TrigPendingF64 {
    emei_d2 e;  // {sin, cos}(k 2pi/256), in flight until end()
    double sr, cr;
};
struct TrigPendingF32 {
    float s, c;
};
__device__ __forceinline__ TrigPendingF64 sincos_begin_ctx(const TrigCtx& t, double x_any) {
    const double inv_step = 40.74366543152521;                                    // 256 / (2 pi)
    const double H1 = 1.5707963267948966 / 64, H2 = 6.123233995736766e-17 / 64;  // 2pi/256 = H1 + H2
    auto begin = [&](double x) __attribute__((always_inline)) {
        TrigPendingF64 p;
        // nearest integer of x * inv_step without v_rndne + v_cvt: adding 1.5 * 2^52 leaves it in the low mantissa
        // bits (|x * inv_step| < 2^31 is guaranteed by the |x| <= 1e6 range of the fast path)
        const double magic = 6755399441055744.0;
        const double shifted = __builtin_fma(x, inv_step, magic);
        const double n = shifted - magic;
        // table index from the low mantissa bits: (k & 255) * 16 + base in two instructions, written out because hipcc turns
        // the C expression into three (shift, mask with 0xff0, add)
        static_assert(kTrigTableSize == 256, "the index mask below");
        uint32_t addr;
        asm("v_and_b32 %0, 0xff, %1\n\tv_lshl_add_u32 %0, %0, 4, %2" : "=v"(addr) : "v"((uint32_t)__double2loint(shifted)), "s"(t.lds_base));
        asm volatile("ds_read_b128 %0, %1" : "=v"(p.e) : "v"(addr));
        // x lives on (it is the state's angle): as fma(-n, H1, x) hipcc copies it first (v_mov_b64 + the two-address v_fmac_f64);
        // the three-address form leaves it where it is
        double r = fma_vsv(n, -H1, x);
        r = __builtin_fma(-n, H2, r);
        const double z = r * r;
        p.sr = __builtin_fma(r * z, fma_vsv(z, 1.0 / 120, t.c3), r);
        p.cr = __builtin_fma(z, __builtin_fma(z, fma_vsv(z, -1.0 / 720, t.c4), -0.5), 1.0);
        return p;
    };
    // Arguments beyond the fast range are repaired AFTER the fact, in a cold, wave-uniform branch that reduces them and runs
    // the block again (for the other lanes of such a wave the second run repeats the first bit for bit): reduced up front and
    // joined in front of one shared copy of the block, the reduced angle and the state's angle need one register, i.e. a
    // v_mov_b64 of the angle on the hot path of every substep.  The first run of a cold lane reads a valid table slot (the
    // index is masked) and its garbage is dropped.
    TrigPendingF64 p = begin(x_any);
    const bool cold = !(__builtin_fabs(x_any) <= kFastTrigLimitF64);
    if (__builtin_expect(__ballot(cold) != 0ull, 0)) {  // wave-uniform entry: see sincos_repair_r
        // hipcc takes an asm's output for available — and, once dead, for FREE — at once: it would hand the first read's
        // destination registers to the reduction below while the LDS is still to deliver into them, and the copies it places
        // at the join would read the second entry before it has arrived.  Both reads are waited for here (cold path only);
        // sincos_end_ctx's own wait then finds nothing pending.
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p.e));
        const double red = trig_reduce_large(x_any);    // straight-line arithmetic: every lane computes it, cold lanes take it
        p = begin(cold ? red : x_any);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p.e));
    }
    return p;
}
__device__ __forceinline__ void sincos_pin(const TrigPendingF64& p, double& a, double& b) {
    asm volatile("" : "+v"(a), "+v"(b) : "v"(p.e));
}
__device__ __forceinline__ void sincos_end_ctx(TrigPendingF64& p, double after0, double after1, double& s, double& c) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p.e) : "v"(after0), "v"(after1));
    s = __builtin_fma(p.e.x, p.cr, p.e.y * p.sr);
    c = __builtin_fma(p.e.y, p.cr, -(p.e.x * p.sr));
}
__device__ __forceinline__ TrigPendingF32 sincos_begin_ctx(const TrigCtx&, float x) {
    TrigPendingF32 p;
    fast_sincosf(x, p.s, p.c);
    return p;
}
__device__ __forceinline__ void sincos_pin(const TrigPendingF32&, float&, float&) {}
__device__ __forceinline__ void sincos_end_ctx(TrigPendingF32& p, float, float, float& s, float& c) { s = p.s, c = p.c; }

__device__ __forceinline__ void sincos_ctx(const TrigCtx& t, double x, double& s, double& c) {
    sincos_fast_ctx(t, trig_arg(x), s, c);
}
// after sincos_end_ctx: float32 still repairs out-of-range arguments after the fact; float64 reduced them up front
__device__ __forceinline__ void sincos_post_ctx(double, double&, double&) {}
__device__ __forceinline__ void sincos_post_ctx(float x, float& s, float& c) { sincos_repair_r(x, s, c); }
__device__ __forceinline__ void sincos_ctx(const TrigCtx& t, float x, float& s, float& c) {
    sincos_fast_ctx(t, x, s, c);
    sincos_repair_r(x, s, c);
}
// every thread of the 256-thread block copies one entry; the barrier must be reached by ALL threads
// (callers stage the table before any early return)
// `rot_c`, `rot_s`: an env whose dynamics only ever use m sin(x + off) and m cos(x + off) stages the table rotated by
// its constant offset and pre-multiplied (InvPend: the pole's mass moment and the angle of its centre of mass at theta = 0;
// rot_c = m cos(off), rot_s = m sin(off)) — the substep then looks its state's angle up as it is and saves the addition
// and the two products; (1, 0) = the plain table, bit for bit.
template <int BLOCK = kBlock>
__device__ __forceinline__ void stage_trig_table(SinCosEntry* lds, const SinCosEntry* __restrict__ src, double rot_c = 1.0, double rot_s = 0.0) {
    static_assert(kTrigTableSize % BLOCK == 0, "every thread copies the same number of entries");
#pragma unroll
    for (int k = 0; k < kTrigTableSize / BLOCK; ++k) {
        SinCosEntry e = src[k * BLOCK + threadIdx.x];
        if (rot_s != 0.0) {
            const double sn = __builtin_fma(e.s, rot_c, e.c * rot_s), cs = __builtin_fma(e.c, rot_c, -(e.s * rot_s));
            e.s = sn, e.c = cs;
        } else if (rot_c != 1.0) {
            e.s *= rot_c, e.c *= rot_c;
        }
        lds[k * BLOCK + threadIdx.x] = e;
    }
    __syncthreads();
}

// hardware reciprocal seed + Newton (emei_math.h)
__device__ __forceinline__ double rcp_r(double d) { return refine_rcp(d, __builtin_amdgcn_rcp(d)); }
__device__ __forceinline__ float rcp_r(float d) { return 1.0f / d; }

// hardware seed (measured 2^-24.4 relative, tools/rcp_accuracy.hip) + ONE Newton step: <= 2.2e-15 relative.  For the
// MuJoCo-backed dynamics (kernel <-> oracle 1e-9); the reference-pinned CartPole keeps the two-step rcp_r.
__device__ __forceinline__ double rcp1_r(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    return __builtin_fma(r0, __builtin_fma(-d, r0, 1.0), r0);
}
__device__ __forceinline__ float rcp1_r(float d) { return 1.0f / d; }
// 1 / sqrt(x): hardware seed + two Newton steps (x = 0 -> +inf, as the seed)
__device__ __forceinline__ double rsqrt_r(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    r = __builtin_fma(r, __builtin_fma(-hx * r, r, 0.5), r);
    return __builtin_fma(r, __builtin_fma(-hx * r, r, 0.5), r);
}
__device__ __forceinline__ float rsqrt_r(float x) { return 1.0f / sqrtf(x); }
// v with the sign bit flipped where `mask` (0 or 0x80000000) says so: J * v for J = +-1 in one 32-bit instruction
__device__ __forceinline__ double flip_sign(double v, uint32_t mask) {
    return __hiloint2double(__double2hiint(v) ^ (int)mask, __double2loint(v));
}
__device__ __forceinline__ float flip_sign(float v, uint32_t mask) { return __uint_as_float(__float_as_uint(v) ^ mask); }

__device__ __forceinline__ double abs_r(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float abs_r(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double copysign_r(double mag, double sgn) { return __builtin_copysign(mag, sgn); }
__device__ __forceinline__ float copysign_r(float mag, float sgn) { return __builtin_copysignf(mag, sgn); }
__device__ __forceinline__ double fmax_r(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float fmax_r(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// n / d for O(1) denominators
__device__ __forceinline__ double div_r(double n, double d) { return n * rcp_r(d); }  // <= ~2 ulp
__device__ __forceinline__ float div_r(float n, float d) { return n / d; }

// Python / NumPy floored modulo a % p for p > 0 (np.remainder), without the library fmod loop:
// k = floor(a/p) from a reciprocal product, remainder by one FMA (exact whenever k is the true
// quotient, because the exact remainder is representable), one-period fix-up when k is off by one.
template <typename T>
__device__ __forceinline__ T pymod_pos(T a, T p, T inv_p) {
    T k = floor(a * inv_p);
    T w = fma_r(-k, p, a);
    if (w < T(0)) w += p;
    else if (w >= p) w -= p;
    return w;
}

// pymod_pos(theta + pi, 2 pi) - pi, the angle wrap of the observations (inverted_pendulum.py:45-49), with the one-period fix-up
// behind ONE wave-uniform test: w in [0, 2 pi) <=> |w - pi| <= pi, so the result of the usual case is tested itself (|o| >= pi:
// the fix-up, and o = -pi exactly, which the cold path leaves alone) — 6 vector instructions instead of 9 per env-step of the
// staged InvertedPendulum kernel, the same bits in every case (the cold path is pymod_pos's own sequence).
template <typename T>
__device__ __forceinline__ T wrap_pi(T theta, bool& finite) {  // finite: of the result (and so of theta); free in the usual case
    const T pi = T(3.141592653589793), p = T(2) * pi;
    const T a = theta + pi;
    const T k = floor(a * T(1.0 / (2 * 3.141592653589793)));
    const T w = fma_r(-k, p, a);
    T o = w - pi;
    finite = true;
    if (__builtin_expect(__ballot(!(fabs(o) < pi)) != 0ull, 0)) {  // also entered by NaN lanes, which it leaves as they are
        T w2 = w;
        if (w2 < T(0)) w2 += p;
        else if (w2 >= p) w2 -= p;
        o = w2 - pi;
        finite = ::isfinite(o);
    }
    return o;
}
template <typename T>
__device__ __forceinline__ T wrap_pi(T theta) {
    bool f;
    return wrap_pi(theta, f);
}

template <typename T>
__device__ __forceinline__ bool finite_r(T v) {
    return ::isfinite(v);
}


// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11): the device-side reset generator.
// key = 64-bit seed; counter = (env lo, env hi, episode, block).  Restated in oracle/emei_oracle.c.
struct u32x4 {
    uint32_t v[4];
};

// KEYS_IN_PLACE (the seed must be wave-uniform, e.g. a kernel argument): the key schedule is bumped by scalar additions the
// compiler cannot hoist.  Left to itself hipcc precomputes the 20 round keys once per kernel; in a kernel whose scalar
// registers are full (the float64 InvertedPendulum) they live in the lanes of a spill register and every redraw of spare
// initial states reads them back with 20 v_readlane_b32.
template <bool KEYS_IN_PLACE = false>
__device__ __forceinline__ u32x4 philox4x32_10(uint64_t seed, uint64_t env, uint32_t episode, uint32_t block) {
    uint32_t c0 = (uint32_t)env, c1 = (uint32_t)(env >> 32), c2 = episode, c3 = block;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (KEYS_IN_PLACE) k0 = __builtin_amdgcn_readfirstlane(k0), k1 = __builtin_amdgcn_readfirstlane(k1);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one full 32x32->64 product per multiplier (v_mad_u64_u32) instead of mul_hi + mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0, c1 = (uint32_t)p1, c2 = n2, c3 = (uint32_t)p0;
        if (KEYS_IN_PLACE) {
            asm("s_add_u32 %0, %0, 0x9E3779B9\n\ts_add_u32 %1, %1, 0xBB67AE85" : "+s"(k0), "+s"(k1) : : "scc");
        } else {
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
    }
    return u32x4{{c0, c1, c2, c3}};
}

__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 0x1.0p-24f; }  // [0,1)

// Standard normals from two Philox words.  Specification (oracle/integrators.h): u1 = ((a >> 8) + 1) / 2^24 in (0, 1], the
// angle t = (b >> 8) / 2^24 TURNS, z = sqrt(-2 ln u1) (cos, sin)(2 pi t).  The hardware float32 transcendentals on these
// exact inputs (v_sin_f32 / v_cos_f32 take turns; v_log_f32 is log2) are within 1.3e-7 abs (sin, cos) and 4.9e-7 abs
// (radius) of the exact values over ALL 2^24 inputs — tools/bm_accuracy.hip; libm's logf / sqrtf / sincosf on the rounded
// angle measured 4.2e-7 / 6.0e-7 — and cost 6 instructions instead of ~170 (sincosf alone brings its large-argument
// reduction into every kernel that can reset an env).
__device__ __forceinline__ void boxmuller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 0x1.0p-24f;  // (0,1], exact
    const float t = (float)(b >> 8) * 0x1.0p-24f;            // [0,1) turns, exact
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln u1 = -2 ln 2 log2 u1
    z0 = rad * __builtin_amdgcn_cosf(t);
    z1 = rad * __builtin_amdgcn_sinf(t);
}

// ---------------------------------------------------------------------------------------------
// action loads: dtype is wave-uniform, so the switch is a scalar branch
__device__ __forceinline__ int load_discrete_action(const void* p, int dtype, int64_t idx) {
    switch (dtype) {
        case EMEI_ACT_U8: return (int)((const uint8_t*)p)[idx];
        case EMEI_ACT_I32: return ((const int32_t*)p)[idx];
        case EMEI_ACT_I64: return (int)((const int64_t*)p)[idx];
        default: return (int)((const float*)p)[idx];
    }
}

}  // namespace emei
