// t1d_kernels.hpp -- the HIP kernels of libt1d_hip.so (gfx950 only).  Included by t1d_abi.hip, which holds the host
// side of the C ABI (include/t1d.h); the per-lane arithmetic (RHS, integrators, sensor, risk) is in t1d_device.hpp.
//
//   step1d_kernel      the headline launch: ONE simulated minute per env.step, split integrator with per-minute step
//                      sizes.  One persistent workgroup per CU whose waves draw 64-env chunks from a queue in LDS; the
//                      main pass integrates the lanes of level 1 and sets those of level 2 (~0.7 %) aside in a list in
//                      LDS, which the waves work off -- 64 at a time, every lane at level 2 -- once the queue is empty.
//   step1_kernel       the same launch with every lane integrated in place: the fixed-step form (level 1 everywhere),
//                      tables of more than 32 patients, batches whose list would not fit in LDS.
//   stepn_kernel       a step of several minutes (sample_time > 1) or one closed-loop step with the controller fused, in
//                      the same persistent form: the state stays in registers across the minutes, a lane that the rule
//                      puts at level 2 is parked in LDS with its state and finished by the pass over the parked lanes.
//   step_kernel        one launch per env.step, any minutes / layout / integrator: pump -> [meal bookkeeping ->
//                      integration -> Gsub -> CGM sample/hold] x minutes -> risk/reward/done. (env.py:48-117)
//   refill_kernel      rebuilds due 150-minute CGM noise blocks ahead of a step kernel compiled without that code.
//   rollout_pid_kernel n_steps x (PID or basal-bolus policy + step) with state in registers.
//   reset_kernel       masked T1DSimEnv.reset().                                              (env.py:119-155)
//   random_meals_kernel  RandomScenario.create_scenario for the whole batch.          (scenario_gen.py:33-60)
//   outcome_kernel     time in range, CVGA percentiles and risk means of a BG history.   (analysis/report.py:74-217)
//   philox_normals_kernel  replays the Philox stream for tests.
#pragma once
#include "../../include/t1d.h"
#include "t1d_device.hpp"

#ifndef T1D_WAVES
#define T1D_WAVES 2
#endif
#ifndef T1D_ROW_RECOMPUTE
#define T1D_ROW_RECOMPUTE 1
#endif
// Tuning builds only (-DT1D_AB_FLAGS=1): bits 0x100 / 0x200 / 0x400 / 0x800 of t1d_batch.flags switch the risk index,
// the pump, the CGM noise and the integration off, to time the rest.  The shipped library is built without them and
// t1d_step rejects unknown flag bits.
// internal bit of KArgs.flags (never in t1d_batch.flags): this launch walks every CU's chunks from the last to the first.
// A launch of 1 Mi fp64 envs touches ~215 MB, about what the 256 MB Infinity Cache in front of HBM holds: walked in the same
// order every launch that is the access pattern an LRU cache serves worst (each line is evicted just before it is wanted
// again); back and forth, a launch starts on what the one before it touched last.
constexpr int kFlagReverse = 0x10000;
#ifndef T1D_AB_FLAGS
#define T1D_AB_FLAGS 0
#endif

#include <climits>
#include <cstdint>
#include <type_traits>

namespace t1d {

template <typename T> struct KArgs {
    int64_t n, env_offset;
    uint64_t seed;
    T* x; T* planned; T* last_qsto; T* last_food; int32_t* t; uint32_t* meta; uint32_t* episode; int32_t* next_meal;
    T* last_cgm; T* ar_e; T* pts; T* prev_risk; T* dbar;
    const T* basal; const T* bolus; const T* cho; const int32_t* meal_time; const T* meal_amt;
    const T* normals; const T* x0_override;
    T* cgm; T* bg; T* reward; uint8_t* done; T* lbgi; T* hbgi; T* risk; T* meal; T* insulin; T* cgm0;
    const T* dpar;          // [DP_COUNT][kMaxPatients] derived patient constants
    const T* prop;          // [prop_rows][np_pad] insulin propagator of the split integrator (kPropRows(n_sub) rows)
    const double* x0tab;    // [13][np]
    const T* minv;          // [11][11] knot second derivatives of the noise spline: M = minv . y
    int* status;
    long long* trace;       // T1D_S1_TRACE builds only: phase timestamps of the first blocks' waves
    SensorC<T> sen; PumpC<T> pump;
    int np, S, n_meals, n_normals, minutes, n_sub, flags, prop_rows, np_pad;
};

template <typename T> struct PidArgs {
    T P, I, D, target;
    T* integ; T* prev; T* sum_risk; T* min_bg; T* max_bg; int32_t* n_low; int32_t* n_high;
    int n_steps;
    int kind;               // 0 = PIDController, 1 = BBController
    const T* bb_basal; const T* bb_cr; const T* bb_cf; T* bb_prev_meal;
    T* bg_trace; T* cgm_trace; T* cho_trace; T* ins_trace; int64_t trace_row;
};

// Row k of a [K][n] array as a wave-uniform base pointer: the lane index i then rides in ONE 32-bit
// VGPR offset shared by every array (global_load ... v_off, s[base]) instead of a 64-bit address
// pair per array kept alive from the first load to the last store.
// The empty asm pins the row base in an SGPR pair and hides its provenance, so loads and stores take the
// `global_* v_off, s[base:base+1]` form and no per-row 64-bit VGPR address survives from load to store.
template <typename U> __device__ __forceinline__ U* row(U* base, int64_t n, int k)
{
#if T1D_ROW_RECOMPUTE
    // the (volatile) asm keeps the row offset in scalar registers AND stops the compiler from hoisting dozens of
    // loop-invariant row pointers out of a tile loop, where they would overflow the SGPR file and be parked in
    // VGPR lanes (v_writelane / v_readlane around every access): a few scalar ops per access are cheaper
    int kk = k;
    asm volatile("" : "+s"(kk));
    return base + (int64_t)kk * n;
#else
    U* p = base + (int64_t)k * n;
    asm volatile("" : "+s"(p));
    return p;
#endif
}
// row whose index may differ between lanes (meal cursor, noise block): ordinary per-lane address
template <typename U> __device__ __forceinline__ U* rowv(U* base, int64_t n, int k) { return base + (int64_t)k * n; }
template <typename A> __device__ __forceinline__ bool ab_flag(const A& a, int bit) { return T1D_AB_FLAGS && (a.flags & bit) != 0; }
// element i of a uniform-base array through an explicit 32-bit BYTE offset (i < 2^28 by contract)
// The access goes through an explicit address_space(1) pointer: row() hides a pointer's provenance, and a
// pointer the compiler cannot trace back to a kernel argument is accessed with FLAT instructions, which
// count on BOTH vmcnt and lgkmcnt -- every LDS wait of the integration loop would then also wait for them.
template <typename U> struct GRef {
    __attribute__((address_space(1))) U* p;
    __device__ __forceinline__ operator U() const { return *p; }
    __device__ __forceinline__ const GRef& operator=(U v) const { *p = v; return *this; }
};
template <typename U> __device__ __forceinline__ GRef<U> at(U* base, unsigned i)
{
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) U gU;
    return GRef<U>{(gU*)((gchar*)base + (unsigned)(i * (unsigned)sizeof(U)))};
}

// Rows of the PACKED state buffers through a buffer resource (single-minute kernels): `buffer_load v, voff, s[rsrc], soff`
// takes the row as one 32-bit scalar byte offset (one s_mul_i32, or a hoisted SGPR) where a 64-bit row pointer costs
// eleven scalar instructions to rebuild -- a third of all instructions the kernel issued -- or two SGPRs to keep, for
// 47 rows.  The resource covers the whole buffer (< 4 GiB: t1d_step), so a lane beyond it reads 0 and writes nothing.
typedef unsigned int t1d_v2u32 __attribute__((ext_vector_type(2)));
template <typename U> struct BRow { __amdgpu_buffer_rsrc_t r; int soff; };
template <typename U> struct BRef {
    __amdgpu_buffer_rsrc_t r; unsigned voff; int soff;
    __device__ __forceinline__ operator U() const
    {
        if constexpr (sizeof(U) == 8) return __builtin_bit_cast(U, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
        else return __builtin_bit_cast(U, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
    __device__ __forceinline__ const BRef& operator=(U v) const
    {
        if constexpr (sizeof(U) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(t1d_v2u32, v), r, voff, soff, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
        return *this;
    }
};
template <typename U> struct BRows {
    __amdgpu_buffer_rsrc_t r; int rowbytes;
    __device__ __forceinline__ BRows(U* base, int64_t n, int rows)
        : r(__builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(rows * n * (int64_t)sizeof(U)), 0x00020000)), rowbytes((int)(n * (int64_t)sizeof(U))) {}
    __device__ __forceinline__ BRow<U> operator()(int k) const { return BRow<U>{r, k * rowbytes}; }
};
template <typename U> __device__ __forceinline__ BRef<U> at(BRow<U> row, unsigned i) { return BRef<U>{row.r, (unsigned)(i * (unsigned)sizeof(U)), row.soff}; }

// env state held in registers across the minutes of a launch
template <typename T> struct Env {
    T x[13];
    T planned, lq, lf, last_cgm, prev_risk;   // prev_risk: risk index of CGM_hist[-1]
    T cur[4];           // current 15-min interval of the noise spline (pts rows 22..25)
    int t, cursor, next_meal, next_meal_loaded;
    bool eating;
};
template <typename T> struct StepOut { T cgm, bg, meal, ins; };

template <typename T>
__device__ __forceinline__ void stage_pars(const KArgs<T>& a, T* lds, int rows = DP_COUNT)
{
    const int tot = rows * kMaxPatients;
    for (int j = threadIdx.x; j < tot; j += blockDim.x) lds[j] = a.dpar[j];
    __syncthreads();
}

// the split integrator's propagator table into (dynamic) LDS; same [rows][np_pad] layout as in memory
template <typename T>
__device__ __forceinline__ void stage_prop(const KArgs<T>& a, T* lds)
{
    const int tot = a.prop_rows * a.np_pad;
    for (int j = threadIdx.x; j < tot; j += blockDim.x) lds[j] = a.prop[j];
    __syncthreads();
}

template <typename T>
__device__ __forceinline__ void load_env(const KArgs<T>& a, unsigned i, uint32_t meta, Env<T>& e)
{
#pragma unroll
    for (int k = 0; k < 13; ++k) e.x[k] = at(row(a.x, a.n, k), i);
    e.planned = at(a.planned, i); e.lq = at(a.last_qsto, i); e.lf = at(a.last_food, i);
    e.last_cgm = at(a.last_cgm, i); e.prev_risk = at(a.prev_risk, i);
#pragma unroll
    for (int k = 0; k < 4; ++k) e.cur[k] = at(row(a.pts, a.n, 22 + k), i);
    e.t = at(a.t, i);
    e.next_meal = a.next_meal ? at(a.next_meal, i) : 0;
    e.next_meal_loaded = e.next_meal;
    e.eating = (meta & T1D_META_EATING) != 0;
    e.cursor = (int)T1D_META_CURSOR(meta);
}

template <typename T>
__device__ __forceinline__ void store_env(const KArgs<T>& a, unsigned i, uint32_t pid, const Env<T>& e)
{
#pragma unroll
    for (int k = 0; k < 13; ++k) at(row(a.x, a.n, k), i) = e.x[k];
    at(a.planned, i) = e.planned; at(a.last_qsto, i) = e.lq; at(a.last_food, i) = e.lf;
    at(a.last_cgm, i) = e.last_cgm; at(a.prev_risk, i) = e.prev_risk;
    if (a.dbar) at(a.dbar, i) = dbar_of(e.lq, e.lf);
    at(a.t, i) = e.t;
    if (a.next_meal && e.next_meal != e.next_meal_loaded) at(a.next_meal, i) = e.next_meal;
    at(a.meta, i) = pid | (e.eating ? T1D_META_EATING : 0u) | (e.planned > T(0) ? T1D_META_PLANNED : 0u) | ((uint32_t)e.cursor << 16);
}

// Refill of the CGM noise deque (noise_gen.py:30-56): ten new AR(1) -> Johnson-SU points at 15-min
// spacing (:84-97) behind the carried-over last point.  Runs once per 150 simulated minutes per env,
// so it is kept out of line: its registers (ocml sinh, Philox) are paid for on this path only.
// Returns sample 0 of the new block, W[0] . points.
// ---- CGM noise (sensor/noise_gen.py) ----------------------------------------------------------------
// The reference interpolates each block of 11 Johnson-SU points (15-min spacing, 150 min) with
// scipy's interp1d(kind='cubic') = the not-a-knot cubic spline, and hands out its values on the sensor
// grid (noise_gen.py:38-47).  That spline is evaluated here in its local form instead of as a dense
// 11-tap operator per sample: with M = second derivatives at the knots (M = Minv . y, Minv fixed),
//   S(tau) = A y_m + B y_{m+1} + ((A^3 - A) M_m + (B^3 - B) M_{m+1}) h^2/6,  A = (t_{m+1} - tau)/h, B = 1 - A,
// so a sample reads 4 words (rows 22..25 of `pts`, the current interval) instead of 11 + 11, and the
// rows it reads do not depend on the env's clock (they can be fetched with the rest of the state).
// pts rows: 0..10 = y (points of the block), 11..21 = M, 22..25 = (y_m, y_{m+1}, M_m, M_{m+1}).
constexpr int kPtsRows = 26;
constexpr int kPackedRows = 18 + kPtsRows + 1;        // rows of the packed state buffer (include/t1d.h): 45 (the last one: dbar)
constexpr int kRowDbar = 18 + kPtsRows;               // 44

#ifndef T1D_REFILL_INLINE
#define T1D_REFILL_INLINE 1
#endif
#if T1D_REFILL_INLINE
#define T1D_REFILL_ATTR __forceinline__
#else
#define T1D_REFILL_ATTR __noinline__
#endif

// Refill of the CGM noise deque (noise_gen.py:30-56): ten new AR(1) -> Johnson-SU points behind the
// carried-over last point (:84-97), then the knot second derivatives.  Once per 150 simulated minutes.
template <typename T>
__device__ T1D_REFILL_ATTR void noise_refill(T* __restrict__ pts, const T* __restrict__ normals, const T* __restrict__ minv,
                                             const uint32_t* __restrict__ episode, int* status, int64_t n, unsigned i,
                                             int64_t env_offset, uint64_t seed, int n_normals, int b, SensorC<T> sen, T* ar_e)
{
    const T p0 = at(rowv(pts, n, b > 0 ? 10 : 0), i);    // carried-over last point (:33,36)
    at(pts, i) = p0;
    T M[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) M[k] = minv[k * 11] * p0;
    T e = *ar_e;
    const uint32_t ep = (!normals && episode) ? at(episode, i) : 0u;
#pragma unroll 1
    for (int q = 0; q < 5; ++q) {
        T z0, z1;
        if (normals) {
            const int d = 1 + 10 * b + 2 * q;
            if (d + 1 < n_normals) { z0 = at(rowv(normals, n, d), i); z1 = at(rowv(normals, n, d + 1), i); }
            else { z0 = z1 = T(0); atomicOr(status, T1D_ST_NORMALS_EXHAUSTED); }
        } else {
            const double2 r = philox_pair(seed, (uint64_t)(env_offset + i), ep, 3u + 5u * (uint32_t)b + (uint32_t)q);
            z0 = (T)r.x; z1 = (T)r.y;
        }
        e = sen.pacf * (e + z0);                         // :88
        const T ya = johnson_su<true>(sen, e);
        e = sen.pacf * (e + z1);
        const T yb = johnson_su<true>(sen, e);
        at(rowv(pts, n, 2 * q + 1), i) = ya;
        at(rowv(pts, n, 2 * q + 2), i) = yb;
        const T* mc = minv + (2 * q + 1);
#pragma unroll
        for (int k = 0; k < 11; ++k) M[k] += mc[k * 11] * ya + mc[k * 11 + 1] * yb;
    }
    *ar_e = e;
#pragma unroll
    for (int k = 0; k < 11; ++k) at(rowv(pts, n, 11 + k), i) = M[k];
}

// next(CGMNoise) for sample index s (noise_gen.py:61-69).  cur = rows 22..25 of pts as loaded with the
// env state; updated (and stored) when the sample enters a new 15-minute interval.
// REFILL = false: the block has already been rebuilt by refill_kernel (t1d_step launches it ahead of the
// step kernel), which keeps the rarely-run refill code -- Philox, Box-Muller, ten Johnson transforms, the
// 11x11 spline operator -- and above all its registers out of the step kernel: with it inlined the fp64
// step kernel needs 256 VGPRs + scratch, without it 180-220 and no scratch (117 vs 140 us at 1 Mi envs).
// x / d and x % d for x >= 0 and a wave-uniform d.  An integer division by a run-time divisor costs ~20 VALU
// instructions, three of them quarter-rate; the sensor grid only ever divides by sample_time (1, 3, 5 minutes
// for the reference's sensors) and by samples-per-block 150 / sample_time, so those take a uniform branch to a
// compile-time divisor (a multiply-high and a shift) and anything else the general path.
__device__ __forceinline__ void divmod_uniform(int x, int d, int& q, int& r)
{
    switch (d) {
        case 1: q = x; r = 0; return;
        case 3: q = x / 3; break;
        case 5: q = x / 5; break;
        case 30: q = x / 30; break;
        case 50: q = x / 50; break;
        case 150: q = x / 150; break;
        default: q = x / d; break;
    }
    r = x - q * d;
}

// entered != nullptr: the caller stores the current interval (rows 22..25) itself where *entered comes back true
template <bool REFILL, typename T>
__device__ __forceinline__ T noise_sample(const KArgs<T>& a, unsigned i, int s, T (&cur)[4], bool* entered = nullptr)
{
    const int64_t n = a.n;
    const int st = a.sen.st;
    int j, b;
    divmod_uniform(s, a.S, b, j);
    const int tau = (j + 1) * st;
    const int m = tau / 15 < 9 ? tau / 15 : 9;
    const int mprev = (tau - st) / 15 < 9 ? (tau - st) / 15 : 9;
    if (REFILL && j == 0) {             // deque empty: build the next 150-minute block
        T e = at(a.ar_e, i);            // the AR(1) state is touched by refills only
        noise_refill<T>(a.pts, a.normals, a.minv, a.episode, a.status, n, i, a.env_offset, a.seed, a.n_normals, b, a.sen, &e);
        at(a.ar_e, i) = e;
    }
    if (j == 0 || m != mprev) {         // entering interval m: fetch its knots (every 15 minutes)
        cur[0] = at(rowv(a.pts, n, m), i);      cur[1] = at(rowv(a.pts, n, m + 1), i);
        cur[2] = at(rowv(a.pts, n, 11 + m), i); cur[3] = at(rowv(a.pts, n, 12 + m), i);
        if (entered) *entered = true;
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) at(row(a.pts, n, 22 + k), i) = cur[k];
        }
    }
    const T B = T(tau - 15 * m) * T(1.0 / 15.0), A = T(1) - B;
    return A * cur[0] + B * cur[1] + T(37.5) * ((A * A * A - A) * cur[2] + (B * B * B - B) * cur[3]);
}

// CGMSensor.measure (cgm.py:26-36) split in two so that the memory latency of the noise block hides
// under the ODE integration: the noise of the sample due at minute t+1 does not depend on the
// patient state, so it is drawn BEFORE the RK4 sub-steps (sample index = 1 + (t+1)/st: reset used
// #0 and #1) and added to Gsub after them.
template <bool REFILL, typename T>
__device__ __forceinline__ T measure_noise(const KArgs<T>& a, unsigned i, Env<T>& e, bool& due, bool* entered = nullptr)
{
    const int t1 = e.t + 1;
    int q, r;
    divmod_uniform(t1, a.sen.st, q, r);
    due = r == 0;
    return due ? noise_sample<REFILL>(a, i, 1 + q, e.cur, entered) : T(0);
}
template <typename T>
__device__ __forceinline__ T measure_apply(const KArgs<T>& a, Env<T>& e, T gsub, T noise, bool due)
{
    if (due) {
        T cgm = gsub + noise;
        cgm = cgm > a.sen.vmin ? cgm : a.sen.vmin;
        cgm = cgm < a.sen.vmax ? cgm : a.sen.vmax;
        e.last_cgm = cgm;
    }
    return e.last_cgm;
}

// scenario.get_action(time) from the per-env meal table (scenario.py:33-42 / scenario_gen.py:23-31).
// With the `next_meal` state array the common minute costs no table access at all: the minute of the
// next entry travels with the env state and the table is touched only when a meal fires.
// HAS_NEXT: the caller knows that the next_meal array exists (packed layout)
template <typename T, bool HAS_NEXT = false, typename E = Env<T>>
__device__ __forceinline__ T meal_lookup(const KArgs<T>& a, unsigned i, E& e)
{
    T meal = T(0);
    if (HAS_NEXT || a.next_meal) {
        if (e.next_meal <= e.t) {                       // rare: a meal fires (or stale entries are skipped)
            while (e.cursor < a.n_meals) {
                const int mt = at(rowv(a.meal_time, a.n, e.cursor), i);
                if (mt > e.t) { e.next_meal = mt; break; }
                if (mt == e.t) meal = at(rowv(a.meal_amt, a.n, e.cursor), i);
                ++e.cursor;
            }
            if (e.cursor >= a.n_meals) e.next_meal = INT_MAX;
            // the table value is waited for inside this rare branch: at the join the wait would also cover, in every
            // minute of every wave, whatever else is in flight (the previous chunk's stores, the next chunk's loads)
            asm volatile("" : "+v"(meal));
        }
    } else if (e.cursor < a.n_meals) {
        int mt = at(rowv(a.meal_time, a.n, e.cursor), i);
        while (mt < e.t && ++e.cursor < a.n_meals) mt = at(rowv(a.meal_time, a.n, e.cursor), i);
        if (e.cursor < a.n_meals && mt == e.t) {
            meal = at(rowv(a.meal_amt, a.n, e.cursor), i);
            ++e.cursor;
        }
    }
    return meal;
}

// T1DSimEnv.step body (env.py:66-84): `minutes` mini_steps with one action.
// PR = NoProp: classical RK4 on all 13 states; otherwise the split integrator, TIERED: step sizes by the rule of
// t1d_device.hpp (every lane its own level, in place), else level 1 in every minute.
template <int MATH, typename T, typename P, bool REFILL = true, typename PR = NoProp, bool TIERED = false>
__device__ __forceinline__ StepOut<T> step_body(const KArgs<T>& a, P& p, unsigned i, Env<T>& e,
                                                T basal, T bolus, bool has_bolus, PR pr = PR())
{
    T q_basal, q_bolus;
    if ((a.flags & T1D_BATCH_NO_PUMP) || ab_flag(a, 0x200)) { // T1DPatient.step driven directly: insulin = basal + bolus as given
        q_basal = basal; q_bolus = has_bolus ? bolus : T(0);
    } else {
        q_basal = pump_quantise(basal, a.pump.inc_basal, a.pump.min_basal, a.pump.max_basal);   // env.py:51
        q_bolus = a.pump.min_bolus > T(0) ? a.pump.min_bolus : T(0);     // = pump.bolus(0)
        if (has_bolus) q_bolus = pump_quantise(bolus, a.pump.inc_bolus, a.pump.min_bolus, a.pump.max_bolus);   // env.py:52
    }
    const T insulin = q_basal + q_bolus;
    const T div = T(a.minutes), inv_div = T(1) / div;
    StepOut<T> o{T(0), T(0), T(0), T(0)};
    for (int m = 0; m < a.minutes; ++m) {
        const T meal = a.cho ? at(row(a.cho, a.n, m), i) : meal_lookup(a, i, e);      // env.py:50
        bool due;
        const T noise = ab_flag(a, 0x400) ? (due = false, T(0)) : measure_noise<REFILL>(a, i, e, due);
        MinuteIn<T> u = eat_minute<MATH, T>(p, e.x, meal, insulin, e.planned, e.lq, e.lf, e.eating);
        if constexpr (PR::kSplit) p.pin_split(); else p.pin();
        if (!ab_flag(a, 0x800)) {
            if constexpr (PR::kSplit && TIERED) split_minute_tiered(p, pr, u, e.x, a.n_sub);
            else if constexpr (PR::kSplit) split_level<1>(p, pr, u, e.x, a.n_sub);
            else rk4_minute<MATH>(p, u, e.x, a.n_sub);
        }
        e.t += 1;
        const T gsub = MATH == 0 ? e.x[12] / p(DP_VG) : e.x[12] * p(DP_IVG);      // t1dpatient.py:217-218
        const T cgm = measure_apply(a, e, gsub, noise, due);                      // env.py:62
        if (MATH == 0) { o.meal += meal / div; o.ins += insulin / div; o.bg += gsub / div; o.cgm += cgm / div; }   // env.py:78-81
        else { o.meal += meal * inv_div; o.ins += insulin * inv_div; o.bg += gsub * inv_div; o.cgm += cgm * inv_div; }
    }
    if (MATH != 0 && a.minutes == 1) { o.ins = insulin; }   // x * (1/1) is exact already; keeps -0 out
    return o;
}

// The default reward (risk_diff, env.py:27-33) is risk(CGM_hist[-2]) - risk(CGM_hist[-1]): the first term is the
// second term of the previous step, so it travels with the env state (prev_risk) instead of being evaluated again.
template <int MATH, typename T>
__device__ __forceinline__ void write_outputs(const KArgs<T>& a, unsigned i, Env<T>& e, const StepOut<T>& o, T rp)
{
    T l, h, r, rc = T(0);
    if (!ab_flag(a, 0x100)) risk_index1<MATH>(o.cgm, l, h, rc);
    at(a.reward, i) = rp - rc;
    e.prev_risk = rc;
    at(a.cgm, i) = o.cgm; at(a.bg, i) = o.bg;
    at(a.done, i) = (o.bg < T(70) || o.bg > T(350)) ? 1 : 0;  // env.py:103
    if (a.lbgi || a.hbgi || a.risk) {
        risk_index1<MATH>(o.bg, l, h, r);                 // env.py:85
        if (a.lbgi) at(a.lbgi, i) = l;
        if (a.hbgi) at(a.hbgi, i) = h;
        if (a.risk) at(a.risk, i) = r;
    }
    if (a.meal) at(a.meal, i) = o.meal;
    if (a.insulin) at(a.insulin, i) = o.ins;
    if (!(fabs((double)e.x[12]) <= 1.0e300)) atomicOr(a.status, T1D_ST_NONFINITE);
}

// VARIANT 0: reference arithmetic (ocml tanh, IEEE divisions as t1dpatient.py writes them), classical RK4, parameters from LDS
//         3: fast arithmetic, classical RK4 evaluated sub-system by sub-system, parameters gathered once per lane into VGPRs
//         4: fast arithmetic, split integrator at level 1 in every minute, VGPR parameters
//         7: fast arithmetic, split integrator with per-minute step sizes taken in place, parameters from LDS
//         6: as 7 with VGPR parameters (fp32, where they fit; fp64 spills)
template <int VARIANT> struct VariantMath { static constexpr int value = VARIANT == 0 ? 0 : 1; };
template <int VARIANT> struct VariantInfo {
    static_assert(VARIANT == 0 || VARIANT == 3 || VARIANT == 4 || VARIANT == 6 || VARIANT == 7, "unknown kernel variant");
    static constexpr bool split = VARIANT == 4 || VARIANT == 6 || VARIANT == 7;
    static constexpr bool tiered = VARIANT == 6 || VARIANT == 7;
    static constexpr bool lds_pars = VARIANT == 0 || VARIANT == 7;
};
extern __shared__ __align__(16) unsigned char t1d_dyn_lds[];

template <int VARIANT, typename T, bool REFILL = true>
__global__ __launch_bounds__(kBlock, T1D_WAVES) void step_kernel(const KArgs<T> a)
{
    constexpr int MATH = VariantMath<VARIANT>::value;
    using VI = VariantInfo<VARIANT>;
    constexpr int kParRows = VI::split ? DP_COUNT : DP_RK4_COUNT;
    __shared__ T lds[VI::lds_pars ? kParRows * kMaxPatients : 1];
    if (VI::lds_pars) stage_pars(a, lds, kParRows);
    if (VI::split) stage_prop(a, (T*)t1d_dyn_lds);
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    __builtin_assume(i < (1u << 28));          // host guarantees n <= 2^28: i * sizeof(T) fits a 32-bit voffset
    if ((int64_t)i >= a.n) return;
    const uint32_t meta = at(a.meta, i);
    const uint32_t pid = T1D_META_PID(meta);
    Env<T> e;
    load_env(a, i, meta, e);
    const T basal = at(a.basal, i);
    const T bolus = a.bolus ? at(a.bolus, i) : T(0);
    const T rp = e.prev_risk;
    StepOut<T> o;
    if constexpr (VARIANT == 4 || VARIANT == 6) {
        ParsReg<T> p;
        p.load(a.dpar, (int)pid);
        o = step_body<MATH, T, ParsReg<T>, REFILL, PropLds<T>, VARIANT == 6>(a, p, i, e, basal, bolus, a.bolus != nullptr, PropLds<T>{(const T*)t1d_dyn_lds, a.np_pad, (int)pid});
    } else if constexpr (VARIANT == 7) {
        ParsLds<T> p{lds, (int)pid};
        o = step_body<MATH, T, ParsLds<T>, REFILL, PropLds<T>, true>(a, p, i, e, basal, bolus, a.bolus != nullptr, PropLds<T>{(const T*)t1d_dyn_lds, a.np_pad, (int)pid});
    } else if constexpr (VARIANT == 3) {
        ParsReg<T> p;
        p.load(a.dpar, (int)pid);
        o = step_body<MATH, T, ParsReg<T>, REFILL>(a, p, i, e, basal, bolus, a.bolus != nullptr);
    } else {
        ParsLds<T> p{lds, (int)pid};
        o = step_body<MATH, T, ParsLds<T>, REFILL>(a, p, i, e, basal, bolus, a.bolus != nullptr);
    }
    write_outputs<MATH>(a, i, e, o, rp);
    store_env(a, i, pid, e);
}

// ---- single-minute step, split integrator, persistent blocks -----------------------------------------
// The launch the headline workload makes a million times: one simulated minute per env.step (1-min sensors),
// no noise-block refill due (refill_kernel ran, or the host vouched).  Differences from step_kernel:
//   * blocks are persistent (grid = what is resident) and walk chunks of 64 envs, so the parameter and
//     propagator tables are staged into LDS once per block instead of once per 256 envs, compactly
//     (row stride 32 or 64 patients, a compile-time constant: every table read is a ds_read with an
//     immediate offset);
//   * everything the integration does not need is stored BEFORE it (meal bookkeeping, clock, meal cursor,
//     insulin/meal outputs), so that only the 13 states, the drawn noise and the previous risk are alive
//     across the sub-step loops -- which is what lets three waves share a SIMD (<= 168 VGPRs) where
//     step_kernel needs ~230.
#ifndef T1D_S1_WAVES
#define T1D_S1_WAVES 3
#endif
#ifndef T1D_S1_TRACE
#define T1D_S1_TRACE 0
#endif
// issue priority of a wave outside its integration (0 = leave priorities alone).  The integration is pure arithmetic;
// everything else leads to a chunk's stores and to the next chunk's loads, and a wave held up there leaves the memory
// system idle: those phases go first on the SIMD, the integrations of the other waves fill the slots they leave
// (1 Mi envs fp64: 84.8 -> 83.0 us; rotating the priority among the waves chunk by chunk instead: 85.5).
#ifndef T1D_S1_PHASE_PRIO
#define T1D_S1_PHASE_PRIO 3
#endif
#if T1D_S1_TRACE
// tuning builds: drain every counter and stamp the wall clock (100 MHz) at phase boundaries
#define S1_MARK(m) do { __builtin_amdgcn_s_waitcnt(0x0070); if (tr && (threadIdx.x & 63) == 0 && tk < 6) tr[tk * 8 + (m)] = (long long)wall_clock64(); } while (0)
#else
#define S1_MARK(m) do { } while (0)
#endif
// one workgroup fills a CU: T1D_S1_WAVES waves on each of its 4 SIMDs in fp64, T1D_S1D_WAVES_F32 in fp32 (see s1d_threads)
template <typename T> constexpr int s1_threads();
#ifndef T1D_S1D_WAVES
#define T1D_S1D_WAVES 3
#endif
#ifndef T1D_S1D_WAVES_F32
#define T1D_S1D_WAVES_F32 4
#endif
// step1d_kernel: three waves per SIMD in fp64 (<= 168 VGPRs); the fp32 instantiation fits four (<= 128: 45.8 against 48.3 us
// at 1 Mi envs -- there the vector pipe and the memory floor are level, and the fourth wave buys overlap)
template <typename T> constexpr int s1d_threads() { return 256 * (sizeof(T) == 4 ? T1D_S1D_WAVES_F32 : T1D_S1D_WAVES); }
template <typename T> constexpr int s1_threads() { return 256 * (sizeof(T) == 4 ? T1D_S1D_WAVES_F32 : T1D_S1_WAVES); }
// EXTRA: the optional outputs (lbgi, hbgi, risk, meal, insulin) exist; without them their five pointers and the
// third risk evaluation drop out of the kernel altogether
//
// One env-minute of one lane: everything between the chunk's loads and its last store.  MODE:
//   0  level 1 for every lane (the fixed-step form of the scheme);
//   1  step sizes by the rule, every lane its own level in place (split_minute_tiered);
//   2  main pass of step1d_kernel: the rule is evaluated right after the meal bookkeeping and handed to
//      on_level(level 2?); lanes of level 1 integrate, the others stop there -- before anything of them is stored;
//   3  pass of step1d_kernel over the listed lanes: every lane at level 2.
// what a chunk reads before its integration
template <typename T> struct S1In { uint32_t meta; int t, next_meal; T basal, bolus, planned, lq, lf, x[13]; };

template <typename T>
__device__ __forceinline__ S1In<T> s1_load(const KArgs<T>& a, unsigned i)
{
    const BRows<T> X(a.x, a.n, kPackedRows);
    const BRows<int32_t> I(a.t, a.n, 3);
    S1In<T> in;
    // what the pump, the meal bookkeeping and the step-size rule's gut part need is requested first: loads return in
    // order, so that work starts while the other eleven state rows are still on their way
    in.meta = (uint32_t)(int32_t)at(I(1), i);
    in.t = at(I(0), i);
    in.next_meal = at(I(2), i);
    in.basal = at(a.basal, i);
    in.bolus = a.bolus ? (T)at(a.bolus, i) : T(0);
    const T dbar = at(X(kRowDbar), i);
#pragma unroll
    for (int k = 0; k < 13; ++k) in.x[k] = at(X(k), i);
    // The three meal words only where they are live (the env is eating or has a meal planned: ~5 % of the env-minutes);
    // elsewhere planned = 0 and a minute needs nothing of last_qsto / last_food but Dbar -- (Dbar, 0) stands in for them
    // (eat_minute forms last_qsto + 1000 last_food: exactly Dbar; a meal that starts overwrites both).  16 bytes less to
    // read per env-minute.  These loads issue behind the state's: they need the meta word, which was requested first.
    if ((in.meta & (T1D_META_EATING | T1D_META_PLANNED)) || (T1D_AB_FLAGS && (a.flags & 0x1000))) {
        in.planned = at(X(13), i); in.lq = at(X(14), i); in.lf = at(X(15), i);
    } else {
        in.planned = T(0); in.lq = dbar; in.lf = T(0);
    }
    return in;
}

// What s1_load fetched, kept in LDS for the listed envs (step1d_kernel): pt = [18][kS1DPark] of T, pi = [3][kS1DPark] ints
constexpr int kS1DPark = 128;                                  // listed envs per CU whose inputs wait in LDS (a multiple of 64)
template <typename T>
__device__ __forceinline__ void s1_park(const S1In<T>& in, T* pt, int* pi, int slot)
{
#pragma unroll
    for (int k = 0; k < 13; ++k) pt[k * kS1DPark + slot] = in.x[k];
    pt[13 * kS1DPark + slot] = in.planned; pt[14 * kS1DPark + slot] = in.lq; pt[15 * kS1DPark + slot] = in.lf;
    pt[16 * kS1DPark + slot] = in.basal; pt[17 * kS1DPark + slot] = in.bolus;
    pi[slot] = (int)in.meta; pi[kS1DPark + slot] = in.t; pi[2 * kS1DPark + slot] = in.next_meal;
}
template <typename T>
__device__ __forceinline__ S1In<T> s1_unpark(const T* pt, const int* pi, int slot)
{
    S1In<T> in;
#pragma unroll
    for (int k = 0; k < 13; ++k) in.x[k] = pt[k * kS1DPark + slot];
    in.planned = pt[13 * kS1DPark + slot]; in.lq = pt[14 * kS1DPark + slot]; in.lf = pt[15 * kS1DPark + slot];
    in.basal = pt[16 * kS1DPark + slot]; in.bolus = pt[17 * kS1DPark + slot];
    in.meta = (uint32_t)pi[slot]; in.t = pi[kS1DPark + slot]; in.next_meal = pi[2 * kS1DPark + slot];
    return in;
}

template <bool REG, typename T, int STRIDE, bool EXTRA, int MODE, typename ONLEVEL>
__device__ __forceinline__ void s1_chunk(const KArgs<T>& a, T* ldp, T* lpr, T* lconst, unsigned i, const S1In<T>& in,
                                         ONLEVEL&& on_level, long long* tr, int tk)
{
    constexpr int LEVEL = MODE == 3 ? 2 : 1;                          // the level this pass integrates at (modes 0, 2, 3)
    (void)tr; (void)tk;
    S1_MARK(0);
    const BRows<T> X(a.x, a.n, kPackedRows);                // rows 0-12 x, 13 planned, 14 last_qsto, 15 last_food, 16 last_cgm, 17 prev_risk, 18.. pts
    const BRows<int32_t> I(a.t, a.n, 3);                    // rows t, meta, next_meal
    const uint32_t meta = in.meta;
    Env<T> e;
    e.t = in.t;
    e.next_meal = in.next_meal;
    const T basal = in.basal;
    const T bolus = in.bolus;
    e.planned = in.planned; e.lq = in.lq; e.lf = in.lf;
#pragma unroll
    for (int k = 0; k < 13; ++k) e.x[k] = in.x[k];
    const uint32_t pid = T1D_META_PID(meta);
    const T planned0 = e.planned, lq0 = e.lq, lf0 = e.lf;
    e.next_meal_loaded = e.next_meal;
    e.eating = (meta & T1D_META_EATING) != 0;
    e.cursor = (int)T1D_META_CURSOR(meta);
    S1_MARK(1);
    T q_basal, q_bolus;
    if (a.flags & T1D_BATCH_NO_PUMP) {
        q_basal = basal; q_bolus = a.bolus ? bolus : T(0);
    } else {
        int z = 0;
        asm volatile("" : "+v"(z));                  // opaque index: the reads below stay inside the chunk loop
        const T* lc = lconst + z;
        q_basal = pump_quantise(basal, lc[0], lc[1], lc[2]);                                   // env.py:51
        q_bolus = lc[4] > T(0) ? lc[4] : T(0);
        if (a.bolus) q_bolus = pump_quantise(bolus, lc[3], lc[4], lc[5]);                      // env.py:52
    }
    const T insulin = q_basal + q_bolus;
    T meal;                                                                                    // env.py:50
    if (a.cho) {
        meal = at(a.cho, i);
        asm volatile("" : "+v"(meal));               // waited for on this path only (see meal_lookup)
    } else {
        meal = meal_lookup<T, true>(a, i, e);
    }
    ParsLdsS<T, STRIDE> pl{ldp, (int)pid};
    MinuteIn<T> u = eat_minute<1, T>(pl, e.x, meal, insulin, e.planned, e.lq, e.lf, e.eating);
    T f1 = T(0);
    if (MODE == 2) {
        const TierPre<T> tp = tier_pre(pl, u, e.x);
        f1 = tp.f1;
        on_level(tp.level2);
        if (tp.level2) return;                       // nothing stored: the pass over the list redoes this lane from its loads
    }
    // bookkeeping is final for this minute: store it now -- the meal words (and Dbar beside them) only where they changed:
    // they do while an env is eating, ~3 % of the minutes.  Where they were not loaded (s1_load) any change means a meal has
    // just started, which rewrites all three.
    {
        const bool live0 = (meta & (T1D_META_EATING | T1D_META_PLANNED)) != 0 || (T1D_AB_FLAGS && (a.flags & 0x1000));
        const bool changed = live0 ? (e.planned != planned0 || e.lq != lq0 || e.lf != lf0) : (e.eating || e.planned > T(0));
        if (changed) {
            at(X(13), i) = e.planned; at(X(14), i) = e.lq; at(X(15), i) = e.lf;
            at(X(kRowDbar), i) = dbar_of(e.lq, e.lf);
        }
    }
    at(I(0), i) = e.t + 1;
    if (e.next_meal != e.next_meal_loaded) at(I(2), i) = e.next_meal;
    {   // patient id, eating / planned flags, meal cursor: changes when a meal starts, ends or fires
        const uint32_t meta1 = pid | (e.eating ? T1D_META_EATING : 0u) | (e.planned > T(0) ? T1D_META_PLANNED : 0u) | ((uint32_t)e.cursor << 16);
        if (meta1 != meta) at(I(1), i) = (int32_t)meta1;
    }
    if (EXTRA) {
        if (a.meal) at(a.meal, i) = meal;
        if (a.insulin) at(a.insulin, i) = insulin;
    }
    S1_MARK(2);
#if T1D_S1_PHASE_PRIO
    if (MODE != 3) __builtin_amdgcn_s_setprio(0);    // the integration fills the issue slots the other phases leave
#endif
    if (!ab_flag(a, 0x800)) {
        PropLdsS<T, STRIDE> pr{lpr, (int)pid};
        if (MODE == 1) {
            split_minute_tiered(pl, pr, u, e.x, a.n_sub);
        } else if (REG) {
            ParsReg<T> p;
#pragma unroll
            for (int k = 0; k < (int)(sizeof(kSplitPars) / sizeof(int)); ++k) p.v[kSplitPars[k]] = pl(kSplitPars[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) p.v[kSplitW(LEVEL) + k] = pl(kSplitW(LEVEL) + k);
            p.pin_split();
            split_level<LEVEL, T, ParsReg<T>, decltype(pr), MODE == 2>(p, pr, u, e.x, a.n_sub, f1);
        } else {
            split_level<LEVEL, T, ParsLdsS<T, STRIDE>, decltype(pr), MODE == 2>(pl, pr, u, e.x, a.n_sub, f1);
        }
    }
    S1_MARK(3);
#if T1D_S1_PHASE_PRIO
    __builtin_amdgcn_s_setprio(T1D_S1_PHASE_PRIO);   // what leads to this chunk's stores and the next chunk's loads goes first
#endif
#pragma unroll
    for (int k = 0; k < 13; ++k) at(X(k), i) = e.x[k];
    // the sensor side is fetched only now: nothing of it has to stay in registers across the integration
#pragma unroll
    for (int k = 0; k < 4; ++k) e.cur[k] = at(X(40 + k), i);
    // with a 1-minute sensor every minute takes a fresh sample: the held value is never read
    T last_cgm = a.sen.st == 1 ? T(0) : (T)at(X(16), i);
    const T rp = at(X(17), i);                                // risk index of the previous step's CGM
    S1_MARK(4);
    bool due, entered = false;
    const T noise = measure_noise<false>(a, i, e, due, &entered);       // e.t is still the minute's start: sample for t + 1
    if (entered) {                                             // the sample opened a new 15-minute interval of the noise spline
#pragma unroll
        for (int k = 0; k < 4; ++k) at(X(40 + k), i) = e.cur[k];
    }
    const T gsub = e.x[12] * pl(DP_IVG);                                                       // t1dpatient.py:217-218
    if (due) {                                                                                 // cgm.py:26-36
        T c = gsub + noise;
        int z = 0;
        asm volatile("" : "+v"(z));
        const T vmin = lconst[6 + z], vmax = lconst[7 + z];
        c = c > vmin ? c : vmin;
        c = c < vmax ? c : vmax;
        last_cgm = c;
        if (a.sen.st != 1) at(X(16), i) = c;                  // the zero-order hold is dead state with a 1-minute sensor
    }
    const T cgm_out = last_cgm, bg_out = gsub;
    at(a.cgm, i) = cgm_out; at(a.bg, i) = bg_out;
    if (!(fabs((double)e.x[12]) <= 1.0e300)) atomicOr(a.status, T1D_ST_NONFINITE);
    T l, h, r, rc = T(0);
    if (!ab_flag(a, 0x100)) risk_index1<1>(cgm_out, l, h, rc);
    at(a.reward, i) = rp - rc;                                                                 // env.py:27-33
    at(X(17), i) = rc;
    at(a.done, i) = (bg_out < T(70) || bg_out > T(350)) ? 1 : 0;                               // env.py:103
    if (EXTRA && (a.lbgi || a.hbgi || a.risk)) {
        risk_index1<1>(bg_out, l, h, r);                                                       // env.py:85
        if (a.lbgi) at(a.lbgi, i) = l;
        if (a.hbgi) at(a.hbgi, i) = h;
        if (a.risk) at(a.risk, i) = r;
    }
#if T1D_S1_TRACE
    if (tr && (threadIdx.x & 63) == 0 && tk < 6) tr[tk * 8 + 5] = (long long)wall_clock64();    // epilogue computed, stores issued
#endif
    S1_MARK(6);
}

struct S1NoLevel { __device__ __forceinline__ void operator()(bool) const {} };

// tables of one CU, staged once per launch: ldp = [DP_COUNT][STRIDE], lpr = [prop_rows][STRIDE]
template <typename T, int STRIDE>
__device__ __forceinline__ void s1_stage_tables(const KArgs<T>& a, T* ldp, T* lpr, T* lconst)
{
    for (int j = threadIdx.x; j < DP_COUNT * STRIDE; j += (int)blockDim.x) {
        const int r = j / STRIDE, c = j % STRIDE;
        ldp[j] = c < a.np ? a.dpar[r * kMaxPatients + c] : T(0);
    }
    for (int j = threadIdx.x; j < a.prop_rows * STRIDE; j += (int)blockDim.x) {
        const int r = j / STRIDE, c = j % STRIDE;
        lpr[j] = c < a.np ? a.prop[r * a.np_pad + c] : T(0);
    }
    if (threadIdx.x == 0) {
        lconst[0] = a.pump.inc_basal; lconst[1] = a.pump.min_basal; lconst[2] = a.pump.max_basal;
        lconst[3] = a.pump.inc_bolus; lconst[4] = a.pump.min_bolus; lconst[5] = a.pump.max_bolus;
        lconst[6] = a.sen.vmin; lconst[7] = a.sen.vmax;
    }
}

// every lane in place: TIERED = step sizes by the rule (LDS parameters), else level 1 everywhere (VGPR parameters)
template <typename T, int STRIDE, bool EXTRA, bool TIERED>
__global__ __launch_bounds__(s1_threads<T>(), 1) void step1_kernel(const KArgs<T> a, int nchunks)
{
    // packed state only (t1d_step checks): rows 13.. of the x buffer are planned, last_qsto, last_food, last_cgm,
    // prev_risk and the 26 noise rows; rows 1, 2 of the t buffer are meta and next_meal.  Deriving them from two
    // base pointers instead of reading ten more kernel arguments keeps the scalar registers from spilling.
    T* const ldp = (T*)t1d_dyn_lds;                        // [DP_COUNT][STRIDE]
    T* const lpr = ldp + DP_COUNT * STRIDE;                // [prop_rows][STRIDE]
    __shared__ int queue;
    __shared__ T lconst[8];                              // pump and sensor limits: read from LDS where used, so that they
                                                         // do not sit in (spilled) scalar registers across the whole kernel
    s1_stage_tables<T, STRIDE>(a, ldp, lpr, lconst);
    if (threadIdx.x == 0) queue = 0;
    __syncthreads();
    // This workgroup owns a contiguous run of 64-env chunks; its waves draw them from a queue in LDS.  The
    // SIMD issues oldest-wave-first, so with a fixed share per wave the first wave of a SIMD would race ahead
    // and the last would finish alone (measured: 66 vs 93 us); with the queue the fast wave simply takes more.
    const int per_block = (nchunks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int first = (int)blockIdx.x * per_block;
    const int count = nchunks - first < per_block ? nchunks - first : per_block;
    const unsigned lane = threadIdx.x & 63u;
#if T1D_S1_TRACE
    long long* tr = (a.trace && blockIdx.x < 32) ? a.trace + (blockIdx.x * (s1_threads<T>() / 64) + threadIdx.x / 64) * 64 : nullptr;
#else
    long long* tr = nullptr;
#endif
    int tk = -1;
    for (int it = 0;; ++it) {
        int c = 0;
        if (lane == 0) c = atomicAdd(&queue, 1);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= count) break;                              // wave-uniform
        const unsigned i = (unsigned)(first + c) * 64u + lane;
        __builtin_assume(i < (1u << 28));
        ++tk;
        // (no `continue` for the lanes beyond the batch: they have to stay with their wave for the next draw)
        if ((int64_t)i < a.n) s1_chunk<!TIERED, T, STRIDE, EXTRA, TIERED ? 1 : 0>(a, ldp, lpr, lconst, i, s1_load(a, i), S1NoLevel(), tr, tk);
    }
}

// The single-minute launch with per-minute step sizes.  Taken in place (step1_kernel<.., true>) a wave with one lane of
// level 2 runs the whole minute at level 2: 0.7 % of the env-minutes of random-meal days are at level 2, but a third of
// the waves hold at least one.  Here the main pass integrates the lanes of level 1 with fixed steps (VGPR parameters, no
// scratch) and a lane of level 2 only leaves its env in a list in LDS -- the decision needs nothing but the state at the
// start of the minute and the first stage's kgut x1, which the integration then reuses, so it is taken right after the
// meal bookkeeping, before anything of the lane is stored.  Waves that find the chunk queue empty wait until every chunk
// of the CU is past that point and then take the listed envs 64 at a time, from their loads, every lane at level 2.  The
// list holds every env of the CU's share (t1d_step sizes it), so it cannot overflow; per-lane arithmetic is that of the
// in-place form, lane for lane.
// (Built, measured and not kept in round 2 -- git history, DESIGN.md: a third, coarser level for calm minutes with two
// lists -- the launch is bound by HBM, not arithmetic, and every listed env costs ~25 scattered 64-byte fetches -- and
// the level of an env's NEXT minute left in `meta` by the previous launch, so that the lists exist at launch start;
// the last eighth of the chunks in a pool any CU draws from through counters in device memory once its own run is done
// (one counter: the draws serialise, +30 us; one per XCD: no faster than without); two waves per SIMD with the next
// chunk's loads issued ahead of the integration: 88 us against 83.)
// The list pass too takes its parameters from VGPRs: it ends the launch alone on its SIMDs, where LDS round trips in the
// dependent chains count.
template <typename T, bool EXTRA>
__global__ __launch_bounds__(s1d_threads<T>(), 1) void step1d_kernel(const KArgs<T> a, int nchunks)
{
    constexpr int STRIDE = 32;
    T* const ldp = (T*)t1d_dyn_lds;                        // [DP_COUNT][STRIDE]
    T* const lpr = ldp + DP_COUNT * STRIDE;                // [prop_rows][STRIDE]
    const int per_block = (nchunks + (int)gridDim.x - 1) / (int)gridDim.x;
    const unsigned lane = threadIdx.x & 63u;
    // what the main pass had loaded of the first kS1DPark listed envs: the pass over the list starts from LDS, not from a
    // second, scattered fetch (which sits on the critical path of the CU's last wave)
    T* const park_t = lpr + a.prop_rows * STRIDE;           // [18][kS1DPark]
    int* const park_i = (int*)(park_t + 18 * kS1DPark);     // [3][kS1DPark]
    // [per_block * 64] envs of level 2, as offsets from the workgroup's first env (< 65536: t1d_step)
    uint16_t* const list = (uint16_t*)(park_i + 3 * kS1DPark);
    __shared__ int queue, taken, listed, passed;
    __shared__ T lconst[8];
    s1_stage_tables<T, STRIDE>(a, ldp, lpr, lconst);
    if (threadIdx.x == 0) { queue = 0; taken = 0; listed = 0; passed = 0; }
    __syncthreads();
    const int first = (int)blockIdx.x * per_block;
    const int count = nchunks - first < per_block ? nchunks - first : per_block;
    const unsigned base = (unsigned)first * 64u;           // the workgroup's first env
    long long* tr = nullptr;
#if T1D_S1_TRACE
    // tuning builds: wall clock (100 MHz) of every wave of the first 32 workgroups at the phase boundaries of the launch
    long long* const ph = (a.trace && blockIdx.x < 32) ? a.trace + (blockIdx.x * (s1d_threads<T>() / 64) + threadIdx.x / 64) * 64 : nullptr;
#define S1D_PHASE(k) do { if (ph && lane == 0) ph[k] = (long long)wall_clock64(); } while (0)
#else
#define S1D_PHASE(k) do { } while (0)
#endif
    S1D_PHASE(0);
    auto draw = [&](int* counter) -> int {
        int g = 0;
        if (lane == 0) g = atomicAdd(counter, 1);
        return __builtin_amdgcn_readfirstlane(g);
    };
    auto on_level = [&](unsigned i, const S1In<T>& in) {
        return [&, i](bool level2) {
            if (level2) {
                const int slot = atomicAdd(&listed, 1);
                list[slot] = (uint16_t)(i - base);
                if (slot < kS1DPark) s1_park(in, park_t, park_i, slot);
            }
            // lane 0 of a chunk is always a live env: it reports the chunk past its decision point, after the
            // list entries of the wave (LDS operations of one wave execute in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicAdd(&passed, 1);
        };
    };
    int it = 0;
    for (;; ++it) {
        const int c = draw(&queue);
        if (c >= count) break;                              // wave-uniform
        const unsigned i = (unsigned)(first + ((a.flags & kFlagReverse) ? count - 1 - c : c)) * 64u + lane;
        __builtin_assume(i < (1u << 28));
        if ((int64_t)i < a.n) {
#if T1D_S1_TRACE
            tr = ph ? ph + 16 : nullptr;                    // per-chunk phase marks of the wave's first six chunks
#endif
            const S1In<T> in = s1_load(a, i);
            s1_chunk<true, T, STRIDE, EXTRA, 2>(a, ldp, lpr, lconst, i, in, on_level(i, in), tr, it);
        }
    }
    tr = nullptr;
    S1D_PHASE(1);
    // every chunk of this CU has been drawn; those still in flight may yet add to the list
    while (__hip_atomic_load(&passed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < count) __builtin_amdgcn_s_sleep(4);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_s_setprio(3);                          // the launch ends with this pass: it goes first on its SIMD
    const int total = __hip_atomic_load(&listed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    S1D_PHASE(2);
#if T1D_S1_TRACE
    if (ph && lane == 0) { ph[6] = total; ph[7] = 0; ph[8] = it; }
#endif
    for (;;) {
        const int lo = draw(&taken) * 64;
        if (lo >= total) break;                             // wave-uniform
        if (lo + (int)lane < total) {
            const unsigned i = base + (unsigned)list[lo + (int)lane];
            __builtin_assume(i < (1u << 28));
            // (wave-uniform: a group of 64 entries lies inside the parked range or beyond it)
            const S1In<T> in = lo + 64 <= kS1DPark ? s1_unpark<T>(park_t, park_i, lo + (int)lane) : s1_load(a, i);
            s1_chunk<true, T, STRIDE, EXTRA, 3>(a, ldp, lpr, lconst, i, in, S1NoLevel(), nullptr, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    S1D_PHASE(3);
    S1D_PHASE(4);
#undef S1D_PHASE
}

// ---- a step of several minutes: state in registers across the minutes, lanes of level 2 set aside -----------------------
// The reference's gym entry point runs Dexcom (sample_time 3, envs/simglucose_gym_env.py:24) and T1DSimEnv.step loops
// int(sample_time) mini-steps under one action (simulation/env.py:75-81): one launch per env.step here too.  One persistent
// workgroup per CU, tables in LDS, as step1d_kernel; the waves of a CU work off WORK ITEMS of up to 64 lanes:
//   a chunk     64 consecutive envs from the CU's queue: loaded once, then `minutes` x [meal -> bookkeeping -> rule ->
//               level-1 integration -> Gsub -> sample/hold] with the state in registers, then the step's epilogue;
//   a record    whenever the rule puts a lane at level 2 the lane leaves its state AS IT STOOD AT THE START OF THAT MINUTE
//               (x, bookkeeping, clock, the means so far, the minute's index) in a record in LDS and is done with the item;
//               the rest of its wave carries on at level 1.  A wave that finds the queue empty takes up to 64 records --
//               as soon as there is one -- and finishes those lanes' steps: each lane from its own minute, every lane at
//               its own level in place (LDS parameters; a wave runs a minute at the level of its most refined lane).
// So the chunks -- 99 % of the env-minutes -- run at level 1 with the state updated in place (a lane of a chunk runs every
// minute or leaves for good: one that sat a minute out and came back would make the compiler keep a second copy of the
// state across the sub-step loops), the records are worked off beside the last chunks, not behind them, and a batch of a
// few chunks per CU has the latency of the in-place kernel.
//   redo        a flagged lane that finds no record free leaves nothing but a bit in a map of the CU's envs (nothing of the
//               item has been stored: the epilogue is where stores happen); once everything else is done the waves go
//               through the map and take those envs again from their loads, like records.  Rare (a batch whose envs all
//               start meals in the same minute); `mode` bit 2 sends every env there (tests).
// CTRL: one closed-loop step per launch -- the controller (PIDController.policy, pid_ctrller.py:17-36, or BBController,
// basal_bolus_ctrller.py:64-79) in the chunk's prologue, its state, statistics and history rows in the epilogue;
// t1d_rollout_* makes one such launch per step for large batches (SimObj.simulate, sim_engine.py:29-39).
template <typename T> struct SnLane {
    T x[13];
    T planned, lq, lf, insulin, bg_sum, cgm_sum, meal_sum, last_cgm;
    int t, next_meal, cursor, m;
    unsigned i;
    uint32_t pid;
    int dirty;          // 1: meal words changed, 2: next_meal changed, 4: eating flag / meal cursor changed
    bool eating;
};
constexpr int kSnParkT = 21, kSnParkI = 5;           // words of a record: 21 of T, 5 ints
constexpr int kSnDirtyShift = 9;                     // the three dirty bits ride in bits 9-11 of the record's meta word
constexpr int kSnConst = 20;                         // lconst: pump 0-5, sensor limits 6-7, step sizes of level 1 8-12, of level 2 13-17
constexpr int kSnSpinLimit = 1 << 20;                // polls before a waiting wave gives up and raises T1D_ST_STALL (never, by design)

// the record's minute word doubles as its "written" flag: -1 until the lane that owns the slot has stored everything else
template <typename T>
__device__ __forceinline__ void sn_park(const SnLane<T>& L, T* pt, int* pi, int cap, int slot)
{
#pragma unroll
    for (int k = 0; k < 13; ++k) pt[k * cap + slot] = L.x[k];
    pt[13 * cap + slot] = L.planned; pt[14 * cap + slot] = L.lq; pt[15 * cap + slot] = L.lf; pt[16 * cap + slot] = L.insulin;
    pt[17 * cap + slot] = L.bg_sum; pt[18 * cap + slot] = L.cgm_sum; pt[19 * cap + slot] = L.meal_sum; pt[20 * cap + slot] = L.last_cgm;
    pi[slot] = (int)L.i; pi[cap + slot] = L.t; pi[2 * cap + slot] = L.next_meal;
    pi[3 * cap + slot] = (int)(L.pid | (L.eating ? T1D_META_EATING : 0u) | ((uint32_t)L.dirty << kSnDirtyShift) | ((uint32_t)L.cursor << 16));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(&pi[4 * cap + slot], L.m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the wave that has claimed records [lo, lo + n) waits until their owners have written them: every lane polls its record's
// flag and the wave leaves together (a wave-uniform loop: no lane spins on its own)
__device__ __forceinline__ void sn_wait_records(int* pi, int cap, int lo, int n, unsigned lane, int* status)
{
    for (int spins = 0;; ++spins) {
        int m = 0;
        if ((int)lane < n) m = __hip_atomic_load(&pi[4 * cap + lo + (int)lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__builtin_amdgcn_ballot_w64(m < 0) == 0ull) break;
        if (spins > kSnSpinLimit) { if (lane == 0) atomicOr(status, T1D_ST_STALL); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <typename T>
__device__ __forceinline__ void sn_unpark(SnLane<T>& L, const T* pt, const int* pi, int cap, int slot)
{
#pragma unroll
    for (int k = 0; k < 13; ++k) L.x[k] = pt[k * cap + slot];
    L.planned = pt[13 * cap + slot]; L.lq = pt[14 * cap + slot]; L.lf = pt[15 * cap + slot]; L.insulin = pt[16 * cap + slot];
    L.bg_sum = pt[17 * cap + slot]; L.cgm_sum = pt[18 * cap + slot]; L.meal_sum = pt[19 * cap + slot]; L.last_cgm = pt[20 * cap + slot];
    L.i = (unsigned)pi[slot]; L.t = pi[cap + slot]; L.next_meal = pi[2 * cap + slot]; L.m = pi[4 * cap + slot];
    const uint32_t w = (uint32_t)pi[3 * cap + slot];
    L.pid = T1D_META_PID(w); L.eating = (w & T1D_META_EATING) != 0; L.dirty = (int)((w >> kSnDirtyShift) & 7u); L.cursor = (int)T1D_META_CURSOR(w);
    if (L.m < 0) L.m = 1 << 30;                             // a record that never arrived (T1D_ST_STALL is up): nothing to run
}

// Two waves per SIMD in fp64: this kernel is bound by arithmetic, not by memory latency (three minutes of integration per
// byte moved), and the fp64 pipe is as full with two waves as with three (tools/ubench/fp64_issue.hip: 5.4 against 5.2
// cycles per v_fma_f64); with 256 registers per lane nothing spills in the minute loop.  (Three waves, 168 registers: 43
// scratch accesses per lane and minute, 31 with the lane's eight idle words set down in LDS across the integration --
// measured slower.)  fp32: four.
#ifndef T1D_SN_WAVES
#define T1D_SN_WAVES 2
#endif
#ifndef T1D_SN_WAVES_F32
#define T1D_SN_WAVES_F32 4
#endif
template <typename T> constexpr int sn_threads() { return 256 * (sizeof(T) == 4 ? T1D_SN_WAVES_F32 : T1D_SN_WAVES); }
// mode bit 1: step sizes by the rule (else level 1 in every minute); bit 2: every env through the redo pass (in place);
// bits 8..: that many waiting records go ahead of a wave's next chunk
template <typename T, bool EXTRA, bool CTRL>
__global__ __launch_bounds__(sn_threads<T>(), 1) void stepn_kernel(const KArgs<T> a, const PidArgs<T> c, int nchunks, int park_cap, int mode)
{
    constexpr int STRIDE = 32;
    constexpr int NT = sn_threads<T>();
    T* const ldp = (T*)t1d_dyn_lds;                        // [DP_COUNT][STRIDE]
    T* const lpr = ldp + DP_COUNT * STRIDE;                // [prop_rows][STRIDE]
    T* const park_t = lpr + a.prop_rows * STRIDE;          // [kSnParkT][park_cap]
    int* const park_i = (int*)(park_t + kSnParkT * park_cap);      // [kSnParkI][park_cap]
    const int per_block = (nchunks + (int)gridDim.x - 1) / (int)gridDim.x;
    unsigned long long* const redo = (unsigned long long*)(park_i + kSnParkI * park_cap + ((kSnParkI * park_cap) & 1));   // [per_block]: envs to take again
    __shared__ int queue, taken, parked, passed, redo_any, redo_next;
    __shared__ T lconst[kSnConst];
    s1_stage_tables<T, STRIDE>(a, ldp, lpr, lconst);
    const int first = (int)blockIdx.x * per_block;
    const int count = nchunks - first < per_block ? nchunks - first : per_block;
    const bool tiered = (mode & 1) != 0, all_redo = (mode & 2) != 0;
    const int group_min = mode >> 8;                        // records that go ahead of the next chunk (1..64)
    if (threadIdx.x == 0) {
        queue = 0; taken = 0; parked = 0; passed = all_redo ? count : 0; redo_any = 0; redo_next = 0;
        T hc[5];
        split_step_sizes<T>(a.n_sub, hc);
        for (int k = 0; k < 5; ++k) lconst[8 + k] = hc[k];
        split_step_sizes<T>(2 * a.n_sub, hc);
        for (int k = 0; k < 5; ++k) lconst[13 + k] = hc[k];
    }
    for (int j = threadIdx.x; j < park_cap; j += NT) park_i[4 * park_cap + j] = -1;
    for (int j = threadIdx.x; j < per_block; j += NT) {
        unsigned long long bits = 0ull;
        if (all_redo && j < count) {                        // every env of the chunk that exists
            const int64_t left = a.n - (int64_t)(first + j) * 64;
            bits = left >= 64 ? ~0ull : (left > 0 ? (1ull << left) - 1ull : 0ull);
        }
        redo[j] = bits;
    }
    if (threadIdx.x == 0 && all_redo) redo_any = 1;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const BRows<T> X(a.x, a.n, kPackedRows);                // rows 0-12 x, 13 planned, 14 last_qsto, 15 last_food, 16 last_cgm, 17 prev_risk, 18.. pts
    const BRows<int32_t> I(a.t, a.n, 3);                    // rows t, meta, next_meal
    const T inv_div = T(1) / T(a.minutes);
    const int st = a.sen.st;
    // ---- the pieces of an item
    // a chunk's loads, the controller (CTRL), the pump (env.py:51-52)
    auto load_lane = [&](SnLane<T>& L, unsigned i) {
        // what the pump and the first minute's bookkeeping need is requested first (loads return in order)
        const uint32_t meta = (uint32_t)(int32_t)at(I(1), i);
        L.t = at(I(0), i);
        L.next_meal = at(I(2), i);
        T basal = T(0), bolus = T(0);
        bool has_bolus = a.bolus != nullptr;
        T obs = T(0), k_integ = T(0), k_prev = T(0), bb_basal = T(0), bb_cr = T(1), bb_cf = T(1), prev_meal = T(0);
        if (CTRL) {
            obs = at(a.cgm, i);
            if (c.kind == 1) { bb_basal = at(c.bb_basal, i); bb_cr = at(c.bb_cr, i); bb_cf = at(c.bb_cf, i); prev_meal = at(c.bb_prev_meal, i); }
            else { k_integ = at(c.integ, i); k_prev = at(c.prev, i); }
        } else {
            basal = at(a.basal, i);
            if (has_bolus) bolus = at(a.bolus, i);
        }
        L.planned = at(X(13), i); L.lq = at(X(14), i); L.lf = at(X(15), i);
#pragma unroll
        for (int k = 0; k < 13; ++k) L.x[k] = at(X(k), i);
        L.last_cgm = st == 1 ? T(0) : (T)at(X(16), i);       // with a 1-minute sensor the held value is never read
        L.i = i; L.m = 0; L.dirty = 0;
        L.pid = T1D_META_PID(meta);
        L.eating = (meta & T1D_META_EATING) != 0;
        L.cursor = (int)T1D_META_CURSOR(meta);
        if (CTRL) {
            const T stT = T(st);
            has_bolus = true;
            if (c.kind == 1) {                              // BBController._bb_policy (basal_bolus_ctrller.py:64-79)
                basal = bb_basal;
                if (prev_meal > T(0)) {
                    const T corr = obs > T(150) ? (obs - c.target) / bb_cf : T(0);
                    bolus = ((prev_meal * stT) / bb_cr + corr) / stT;
                }
            } else {                                        // PIDController.policy (pid_ctrller.py:17-36); its state moves on in the epilogue
                basal = c.P * (obs - c.target) + c.I * k_integ + c.D * (obs - k_prev) / stT;
            }
        }
        T q_basal, q_bolus;
        if (a.flags & T1D_BATCH_NO_PUMP) {
            q_basal = basal; q_bolus = has_bolus ? bolus : T(0);
        } else {
            int z = 0;
            asm volatile("" : "+v"(z));
            const T* lc = lconst + z;
            q_basal = pump_quantise(basal, lc[0], lc[1], lc[2]);                               // env.py:51
            q_bolus = lc[4] > T(0) ? lc[4] : T(0);
            if (has_bolus) q_bolus = pump_quantise(bolus, lc[3], lc[4], lc[5]);                // env.py:52
        }
        L.insulin = q_basal + q_bolus;
        L.bg_sum = T(0); L.cgm_sum = T(0); L.meal_sum = T(0);
    };
    // scenario meal (env.py:50) -> meal bookkeeping (t1dpatient.py:82-107) -> the first gastric-emptying flux and the rule
    auto minute_head = [&](SnLane<T>& L, T& meal, TierPre<T>& tp) -> MinuteIn<T> {
        if (a.cho) {
            meal = at(rowv(a.cho, a.n, L.m), L.i);
            asm volatile("" : "+v"(meal));
        } else {
            meal = meal_lookup<T, true>(a, L.i, L);
        }
        ParsLdsS<T, STRIDE> pl{ldp, (int)L.pid};
        MinuteIn<T> u = eat_minute<1, T>(pl, L.x, meal, L.insulin, L.planned, L.lq, L.lf, L.eating);
        if (tiered) tp = tier_pre(pl, u, L.x);
        else { tp.f1 = kgut_flux(pl, u, L.x[0], L.x[1]); tp.level2 = false; }
        return u;
    };
    // the parameters of a chunk's minute loop, gathered into registers once per chunk: 16 model constants + the 4 gut weights
    // of level 1
    auto gather_pars = [&](ParsReg<T>& p, uint32_t pid) {
        ParsLdsS<T, STRIDE> pl{ldp, (int)pid};
#pragma unroll
        for (int k = 0; k < (int)(sizeof(kSplitPars) / sizeof(int)); ++k) p.v[kSplitPars[k]] = pl(kSplitPars[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) p.v[kSplitW(1) + k] = pl(kSplitW(1) + k);
    };
    // ... and for the records: the weights of both levels (a lone wave cannot hide the LDS latency of ~200 parameter reads
    // per minute behind anything: from LDS the pass over the records took three times as long)
    auto gather_pars2 = [&](ParsReg<T>& p, uint32_t pid) {
        gather_pars(p, pid);
        ParsLdsS<T, STRIDE> pl{ldp, (int)pid};
#pragma unroll
        for (int k = 0; k < 4; ++k) p.v[kSplitW(2) + k] = pl(kSplitW(2) + k);
    };
    // one minute of a chunk at level 1: step sizes from LDS
    auto integrate = [&](ParsReg<T>& p, SnLane<T>& L, const MinuteIn<T>& u, T f1) {
        PropLdsS<T, STRIDE> pr{lpr, (int)L.pid};
        p.pin_split();
        split_level<1, T, ParsReg<T>, decltype(pr), true, true>(p, pr, u, L.x, a.n_sub, f1, false, lconst + 8);
    };
    // clock, Gsub (t1dpatient.py:217-218), the CGM sample if one is due (cgm.py:26-36), the step's means (env.py:78-81)
    auto minute_tail = [&](SnLane<T>& L, T meal) {
        L.t += 1;
        ParsLdsS<T, STRIDE> pl{ldp, (int)L.pid};
        const T gsub = L.x[12] * pl(DP_IVG);
        int q, r;
        divmod_uniform(L.t, st, q, r);
        if (r == 0) {
            T cur[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = at(X(40 + k), L.i);
            bool entered = false;
            const T noise = noise_sample<false>(a, L.i, 1 + q, cur, &entered);
            if (entered) {
#pragma unroll
                for (int k = 0; k < 4; ++k) at(X(40 + k), L.i) = cur[k];
            }
            int z = 0;
            asm volatile("" : "+v"(z));
            const T vmin = lconst[6 + z], vmax = lconst[7 + z];
            T cv = gsub + noise;
            cv = cv > vmin ? cv : vmin;
            cv = cv < vmax ? cv : vmax;
            L.last_cgm = cv;
        }
        L.meal_sum += meal * inv_div; L.bg_sum += gsub * inv_div; L.cgm_sum += L.last_cgm * inv_div;
    };
    // the step's epilogue (env.py:85-117) and the state's way back
    auto epilogue = [&](SnLane<T>& L) {
        const unsigned i = L.i;
#pragma unroll
        for (int k = 0; k < 13; ++k) at(X(k), i) = L.x[k];
        if (L.dirty & 1) { at(X(13), i) = L.planned; at(X(14), i) = L.lq; at(X(15), i) = L.lf; at(X(kRowDbar), i) = dbar_of(L.lq, L.lf); }
        at(I(0), i) = L.t;
        if (L.dirty & 2) at(I(2), i) = L.next_meal;
        if (L.dirty & 5) at(I(1), i) = (int32_t)(L.pid | (L.eating ? T1D_META_EATING : 0u) | (L.planned > T(0) ? T1D_META_PLANNED : 0u) | ((uint32_t)L.cursor << 16));
        if (st != 1) at(X(16), i) = L.last_cgm;
        const T rp = at(X(17), i);                          // risk index of the previous step's CGM
        const T cgm_out = L.cgm_sum, bg_out = L.bg_sum;
        if (CTRL && c.kind == 0) {                          // pid_ctrller.py:30-33: the observation the policy acted on is still in place
            const T obs = at(a.cgm, i);
            at(c.integ, i) = (T)at(c.integ, i) + (obs - c.target) * T(st);
            at(c.prev, i) = obs;
        }
        at(a.cgm, i) = cgm_out; at(a.bg, i) = bg_out;
        if (!(fabs((double)L.x[12]) <= 1.0e300)) atomicOr(a.status, T1D_ST_NONFINITE);
        T l, h, r = T(0), rc = T(0);
        risk_index1<1>(cgm_out, l, h, rc);
        at(a.reward, i) = rp - rc;                                                             // env.py:27-33
        at(X(17), i) = rc;
        at(a.done, i) = (bg_out < T(70) || bg_out > T(350)) ? 1 : 0;                           // env.py:103
        T ins_out = L.insulin;
        if (a.minutes != 1) {                               // the mean as env.py:79 accumulates it
            ins_out = T(0);
            for (int m = 0; m < a.minutes; ++m) ins_out += L.insulin * inv_div;
        }
        if (EXTRA) {
            const bool want_risk = a.lbgi || a.hbgi || a.risk || (CTRL && c.sum_risk);
            if (want_risk) risk_index1<1>(bg_out, l, h, r);                                    // env.py:85
            if (a.lbgi) at(a.lbgi, i) = l;
            if (a.hbgi) at(a.hbgi, i) = h;
            if (a.risk) at(a.risk, i) = r;
            if (a.meal) at(a.meal, i) = L.meal_sum;
            if (a.insulin) at(a.insulin, i) = ins_out;
        }
        if (CTRL) {
            if (c.kind == 1) at(c.bb_prev_meal, i) = L.meal_sum;
            if (c.bg_trace) c.bg_trace[c.trace_row * a.n + i] = bg_out;
            if (c.cgm_trace) c.cgm_trace[c.trace_row * a.n + i] = cgm_out;
            if (c.cho_trace) c.cho_trace[c.trace_row * a.n + i] = L.meal_sum;
            if (c.ins_trace) c.ins_trace[c.trace_row * a.n + i] = ins_out;
            if (c.sum_risk) at(c.sum_risk, i) = (T)at(c.sum_risk, i) + r;
            if (c.min_bg) { const T v = at(c.min_bg, i); at(c.min_bg, i) = bg_out < v ? bg_out : v; }
            if (c.max_bg) { const T v = at(c.max_bg, i); at(c.max_bg, i) = bg_out > v ? bg_out : v; }
            if (c.n_low) at(c.n_low, i) = (int)at(c.n_low, i) + (bg_out < T(70) ? 1 : 0);
            if (c.n_high) at(c.n_high, i) = (int)at(c.n_high, i) + (bg_out > T(180) ? 1 : 0);
        }
    };
    for (;;) {
        // ---- the next work item: a chunk while the queue lasts, else up to 64 records -- whatever is there: a wave with
        // nothing else to do takes a single record too (the record's minutes then run beside the last chunks, not behind
        // them) --, at the very end the words of the redo map
        int w_chunk = -1, w_lo = 0, w_n = 0, w_redo = -1;
        if (lane == 0) {
            // a full wave's worth of records goes ahead of the next chunk: worked off in full waves while the chunks last,
            // they leave only the stragglers for the end of the launch
            bool full = false;
            if (!all_redo) {
                int pk = __hip_atomic_load(&parked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pk = pk < park_cap ? pk : park_cap;
                int tk = __hip_atomic_load(&taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int av = pk - tk < 64 ? pk - tk : 64;
                if (av >= group_min && __hip_atomic_compare_exchange_strong(&taken, &tk, tk + av, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    w_lo = tk; w_n = av; full = true;
                }
            }
            const int cc = (all_redo || full) ? count : atomicAdd(&queue, 1);
            if (cc < count) w_chunk = cc;
            else if (!full) {
                for (int spins = 0;; ++spins) {
                    // `passed` first: once every chunk is past its last decision the record count is final
                    const int ps = __hip_atomic_load(&passed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    int pk = __hip_atomic_load(&parked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pk = pk < park_cap ? pk : park_cap;
                    int tk = __hip_atomic_load(&taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int avail = pk - tk;
                    if (avail > 0) {
                        const int n = avail < 64 ? avail : 64;
                        if (__hip_atomic_compare_exchange_strong(&taken, &tk, tk + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            w_lo = tk; w_n = n;
                            break;
                        }
                        continue;
                    }
                    if (ps >= count) {
                        // no record left and no chunk that could still leave one: what remains is the redo map, final now
                        if (__hip_atomic_load(&redo_any, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            const int rw = atomicAdd(&redo_next, 1);
                            if (rw < count) w_redo = rw;
                        }
                        break;
                    }
                    if (spins > kSnSpinLimit) { atomicOr(a.status, T1D_ST_STALL); break; }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
        }
        w_chunk = __builtin_amdgcn_readfirstlane(w_chunk);
        w_lo = __builtin_amdgcn_readfirstlane(w_lo);
        w_n = __builtin_amdgcn_readfirstlane(w_n);
        w_redo = __builtin_amdgcn_readfirstlane(w_redo);
        if (w_chunk < 0 && w_n == 0 && w_redo < 0) break;   // wave-uniform
#if T1D_S1_PHASE_PRIO
        __builtin_amdgcn_s_setprio(T1D_S1_PHASE_PRIO);
#endif
        if (w_chunk < 0) {
            // ---- records: each lane from the minute it was parked in (which the rule, on the same state, puts at level 2
            // again); a word of the redo map: the envs of a chunk that found no record free, from their loads.  Every lane
            // at its own level, in place.
            // (Every lane reads -- the idle ones a record or env of a neighbour, and do nothing with it: a load under a
            // divergent branch writes only the active lanes of its registers, which makes the compiler keep the previous
            // item's values alive in the others across the whole item loop: 27 register pairs spilled and reloaded per item.)
            SnLane<T> L;
            bool active;
            if (w_n > 0) {
                sn_wait_records(park_i, park_cap, w_lo, w_n, lane, a.status);
                active = (int)lane < w_n;
                sn_unpark(L, park_t, park_i, park_cap, w_lo + (active ? (int)lane : 0));
            } else {
                const unsigned i = (unsigned)(first + w_redo) * 64u + lane;
                __builtin_assume(i < (1u << 28));
                active = ((redo[w_redo] >> lane) & 1ull) != 0ull;
                load_lane(L, (int64_t)i < a.n ? i : (unsigned)(a.n - 1));
            }
            // (fp64; the fp32 instantiation has no registers to spare at four waves per SIMD and keeps reading LDS)
            constexpr bool kRecReg = sizeof(T) == 8;
            ParsReg<T> p;
            if constexpr (kRecReg) gather_pars2(p, L.pid);
            for (;;) {
                const bool on = active && L.m < a.minutes;
                if (__builtin_amdgcn_ballot_w64(on) == 0ull) break;             // wave-uniform
                if (on) {
                    T meal = T(0);
                    TierPre<T> tp{T(0), false};
                    const T planned0 = L.planned, lq0 = L.lq, lf0 = L.lf;
                    const int nm0 = L.next_meal, cur0 = L.cursor;
                    const bool eat0 = L.eating;
                    const MinuteIn<T> u = minute_head(L, meal, tp);
                    L.dirty |= ((L.planned != planned0 || L.lq != lq0 || L.lf != lf0) ? 1 : 0) | (L.next_meal != nm0 ? 2 : 0) |
                               ((L.cursor != cur0 || L.eating != eat0) ? 4 : 0);
                    PropLdsS<T, STRIDE> pr{lpr, (int)L.pid};
                    if constexpr (kRecReg) {
                        p.pin_split();
                        split_level<0, T, ParsReg<T>, decltype(pr), true, true>(p, pr, u, L.x, a.n_sub, tp.f1, tp.level2, lconst + 8);
                    } else {
                        ParsLdsS<T, STRIDE> pl{ldp, (int)L.pid};
                        split_level<0, T, ParsLdsS<T, STRIDE>, decltype(pr), true>(pl, pr, u, L.x, a.n_sub, tp.f1, tp.level2);
                    }
                    minute_tail(L, meal);
                    L.m += 1;
                }
            }
            if (active) { epilogue(L); }
            continue;
        }
        // ---- a chunk
        const unsigned i0 = (unsigned)(first + ((a.flags & kFlagReverse) ? count - 1 - w_chunk : w_chunk)) * 64u + lane;   // (kFlagReverse: above)
        __builtin_assume(i0 < (1u << 28));
        bool left = (int64_t)i0 >= a.n;                     // lanes beyond the batch have nothing to finish
        SnLane<T> L;
        {
            load_lane(L, left ? (unsigned)(a.n - 1) : i0);  // (every lane loads: see above)
            ParsReg<T> p;
            gather_pars(p, L.pid);
            while (!left && L.m < a.minutes) {
                T meal = T(0);
                TierPre<T> tp{T(0), false};
                // the lane as it stands at the start of the minute: what a record holds
                const T planned0 = L.planned, lq0 = L.lq, lf0 = L.lf;
                const int nm0 = L.next_meal, cur0 = L.cursor;
                const bool eat0 = L.eating;
                const MinuteIn<T> u = minute_head(L, meal, tp);
                if (tp.level2) {
                    // level 2: leave a record (or, no record free, a bit in the redo map) and be done with this item
                    L.planned = planned0; L.lq = lq0; L.lf = lf0; L.next_meal = nm0; L.cursor = cur0; L.eating = eat0;
                    const int slot = atomicAdd(&parked, 1);
                   
                    if (slot < park_cap) sn_park(L, park_t, park_i, park_cap, slot);
                    else {
                       
                        const unsigned rel = L.i - (unsigned)first * 64u;
                        atomicOr(&redo[rel >> 6], 1ull << (rel & 63u));
                        redo_any = 1;
                    }
                    left = true;
                    break;
                }
                L.dirty |= ((L.planned != planned0 || L.lq != lq0 || L.lf != lf0) ? 1 : 0) | (L.next_meal != nm0 ? 2 : 0) |
                           ((L.cursor != cur0 || L.eating != eat0) ? 4 : 0);
#if T1D_S1_PHASE_PRIO
                __builtin_amdgcn_s_setprio(0);              // the integration fills the issue slots the other phases leave
#endif
                integrate(p, L, u, tp.f1);
#if T1D_S1_PHASE_PRIO
                __builtin_amdgcn_s_setprio(T1D_S1_PHASE_PRIO);
#endif
                minute_tail(L, meal);
                L.m += 1;
            }
        }
        // past this chunk's last decision: it can leave no more records.  (Lane 0 of a chunk is always a live env.  The
        // count goes up BEFORE the epilogue, not as the loop body's last statement: a one-lane atomic right in front of
        // the back edge came out of hipcc 7.2 with the other 63 lanes dropped from the next trip.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) atomicAdd(&passed, 1);
        if (!left) { epilogue(L); }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
}

// Rebuilds the CGM noise block of every env whose next sample(s) -- in minutes (t, t + minutes] -- start a
// new 150-minute block.  Launched by t1d_step ahead of step_kernel<.., REFILL = false>; touches 4 B per env
// (the clock) unless a refill is due, which happens once per 150 simulated minutes per env.
template <typename T>
__global__ __launch_bounds__(kBlock) void refill_kernel(const KArgs<T> a)
{
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    __builtin_assume(i < (1u << 28));
    if ((int64_t)i >= a.n) return;
    const int t = at(a.t, i);
    for (int m = 1; m <= a.minutes; ++m) {
        const int t1 = t + m;
        int q, r, blk, j;
        divmod_uniform(t1, a.sen.st, q, r);
        if (r != 0) continue;
        divmod_uniform(1 + q, a.S, blk, j);
        if (j != 0) continue;
        T e = at(a.ar_e, i);
        noise_refill<T>(a.pts, a.normals, a.minv, a.episode, a.status, a.n, i, a.env_offset, a.seed, a.n_normals, blk, a.sen, &e);
        at(a.ar_e, i) = e;
    }
}

template <int VARIANT, typename T, typename P, typename PR = NoProp>
__device__ __forceinline__ void rollout_body(const KArgs<T>& a, const PidArgs<T>& c, P& p, unsigned i, uint32_t pid, Env<T>& e, PR pr = PR())
{
    constexpr int MATH = VariantMath<VARIANT>::value;
    T obs = at(a.cgm, i);
    const bool bb = c.kind == 1;
    T integ = T(0), prev = T(0), bb_basal = T(0), bb_cr = T(1), bb_cf = T(1), prev_meal = T(0);
    if (bb) { bb_basal = at(c.bb_basal, i); bb_cr = at(c.bb_cr, i); bb_cf = at(c.bb_cf, i); prev_meal = at(c.bb_prev_meal, i); }
    else { integ = at(c.integ, i); prev = at(c.prev, i); }
    T sum_risk = c.sum_risk ? at(c.sum_risk, i) : T(0);
    T min_bg = c.min_bg ? at(c.min_bg, i) : T(0), max_bg = c.max_bg ? at(c.max_bg, i) : T(0);
    int n_low = c.n_low ? at(c.n_low, i) : 0, n_high = c.n_high ? at(c.n_high, i) : 0;
    const T st = T(a.sen.st);
    StepOut<T> o{obs, T(0), T(0), T(0)};
    T cgm_before = T(0);                        // CGM of the step before the last one, once two steps have run
    for (int s = 0; s < c.n_steps; ++s) {
        T u, bolus = T(0);
        if (bb) {
            // BBController._bb_policy (basal_bolus_ctrller.py:64-79)
            u = bb_basal;
            if (prev_meal > T(0)) {
                const T corr = obs > T(150) ? (obs - c.target) / bb_cf : T(0);
                bolus = ((prev_meal * st) / bb_cr + corr) / st;
            }
        } else {
            // PIDController.policy (pid_ctrller.py:17-36)
            u = c.P * (obs - c.target) + c.I * integ + c.D * (obs - prev) / st;
            prev = obs;
            integ += (obs - c.target) * st;
        }
        o = step_body<MATH, T, P, true, PR, VariantInfo<VARIANT>::tiered>(a, p, i, e, u, bolus, true, pr);
        obs = o.cgm;
        prev_meal = o.meal;
        if (c.bg_trace) c.bg_trace[(c.trace_row + s) * a.n + i] = o.bg;
        if (c.cgm_trace) c.cgm_trace[(c.trace_row + s) * a.n + i] = o.cgm;
        if (c.cho_trace) c.cho_trace[(c.trace_row + s) * a.n + i] = o.meal;
        if (c.ins_trace) c.ins_trace[(c.trace_row + s) * a.n + i] = o.ins;
        if (s + 1 < c.n_steps) cgm_before = o.cgm;  // CGM history advances every step
        if (c.sum_risk) { T l, h, r; risk_index1<MATH>(o.bg, l, h, r); sum_risk += r; }
        min_bg = o.bg < min_bg ? o.bg : min_bg;
        max_bg = o.bg > max_bg ? o.bg : max_bg;
        n_low += o.bg < T(70); n_high += o.bg > T(180);
    }
    // the last step's reward: against the step before it, or against what the state carried in (one step)
    T rp = e.prev_risk;
    if (c.n_steps > 1) { T l, h; risk_index1<MATH>(cgm_before, l, h, rp); }
    write_outputs<MATH>(a, i, e, o, rp);
    store_env(a, i, pid, e);
    if (bb) at(c.bb_prev_meal, i) = prev_meal;
    else { at(c.integ, i) = integ; at(c.prev, i) = prev; }
    if (c.sum_risk) at(c.sum_risk, i) = sum_risk;
    if (c.min_bg) at(c.min_bg, i) = min_bg;
    if (c.max_bg) at(c.max_bg, i) = max_bg;
    if (c.n_low) at(c.n_low, i) = n_low;
    if (c.n_high) at(c.n_high, i) = n_high;
}

template <int VARIANT, typename T>
__global__ __launch_bounds__(kBlock, T1D_WAVES) void rollout_pid_kernel(const KArgs<T> a, const PidArgs<T> c)
{
    using VI = VariantInfo<VARIANT>;
    constexpr int kParRows = VI::split ? DP_COUNT : DP_RK4_COUNT;
    __shared__ T lds[VI::lds_pars ? kParRows * kMaxPatients : 1];
    if (VI::lds_pars) stage_pars(a, lds, kParRows);
    if (VI::split) stage_prop(a, (T*)t1d_dyn_lds);
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    __builtin_assume(i < (1u << 28));          // host guarantees n <= 2^28: i * sizeof(T) fits a 32-bit voffset
    if ((int64_t)i >= a.n) return;
    const uint32_t meta = at(a.meta, i);
    const uint32_t pid = T1D_META_PID(meta);
    Env<T> e;
    load_env(a, i, meta, e);
    if constexpr (VARIANT == 4 || VARIANT == 6) {
        ParsReg<T> p;
        p.load(a.dpar, (int)pid);
        rollout_body<VARIANT>(a, c, p, i, pid, e, PropLds<T>{(const T*)t1d_dyn_lds, a.np_pad, (int)pid});
    } else if constexpr (VARIANT == 7) {
        ParsLds<T> p{lds, (int)pid};
        rollout_body<VARIANT>(a, c, p, i, pid, e, PropLds<T>{(const T*)t1d_dyn_lds, a.np_pad, (int)pid});
    } else if constexpr (VARIANT == 3) {
        ParsReg<T> p;
        p.load(a.dpar, (int)pid);
        rollout_body<VARIANT>(a, c, p, i, pid, e);
    } else {
        ParsLds<T> p{lds, (int)pid};
        rollout_body<VARIANT>(a, c, p, i, pid, e);
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void reset_kernel(const KArgs<T> a, const uint8_t* mask, int random_init_bg)
{
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    __builtin_assume(i < (1u << 28));          // host guarantees n <= 2^28: i * sizeof(T) fits a 32-bit voffset
    if ((int64_t)i >= a.n) return;
    if (mask && !at(mask, i)) return;
    const int64_t n = a.n;
    const uint32_t pid = T1D_META_PID(at(a.meta, i));
    uint32_t ep = 0;
    if (a.episode) { ep = at(a.episode, i) + 1u; at(a.episode, i) = ep; }
    const uint64_t gid = (uint64_t)(a.env_offset + i);
    Env<T> e;
    // T1DPatient.reset (t1dpatient.py:247-281)
#pragma unroll
    for (int k = 0; k < 13; ++k)
        e.x[k] = a.x0_override ? at(row(a.x0_override, n, k), i) : (T)a.x0tab[k * a.np + pid];
    if (random_init_bg && !a.x0_override) {          // :256-270, statistical counterpart
        const double2 r1 = philox_pair(a.seed, gid, ep, 1u), r2 = philox_pair(a.seed, gid, ep, 2u);
        e.x[3] += t_sqrt(T(0.1) * e.x[3]) * (T)r1.x;
        e.x[4] += t_sqrt(T(0.1) * e.x[4]) * (T)r1.y;
        e.x[12] += t_sqrt(T(0.1) * e.x[12]) * (T)r2.x;
    }
    e.planned = T(0); e.lq = e.x[0] + e.x[1]; e.lf = T(0); e.eating = false; e.cursor = 0; e.t = 0;
    e.next_meal = (a.n_meals > 0) ? at(a.meal_time, i) : INT_MAX;      // first table row; entries before t = 0 are skipped lazily
    e.next_meal_loaded = e.next_meal - 1;                               // force the store
    // CGMSensor.reset -> CGMNoise(): first AR value and first 15-min point (noise_gen.py:24,86)
    T z0;
    if (a.normals) {
        if (a.n_normals > 0) z0 = at(a.normals, i); else { z0 = T(0); atomicOr(a.status, T1D_ST_NORMALS_EXHAUSTED); }
    } else {
        z0 = (T)philox_pair(a.seed, gid, ep, 0u).x;
    }
    at(a.ar_e, i) = z0;
    at(a.pts, i) = johnson_su<true>(a.sen, z0);
    e.last_cgm = T(0);
    const T vg = a.dpar[DP_VG * kMaxPatients + pid];
    const T bg0 = e.x[12] / vg;
    T c[2];
    for (int s = 0; s < 2; ++s) {                    // env.py:126 (history[0]) and env.py:142 (observation)
        T v = bg0 + noise_sample<true>(a, i, s, e.cur);
        v = v > a.sen.vmin ? v : a.sen.vmin;
        v = v < a.sen.vmax ? v : a.sen.vmax;
        c[s] = v;
    }
    e.last_cgm = c[1];
    { T l0, h0; risk_index1<0>(c[0], l0, h0, e.prev_risk); }    // CGM_hist = [sample #0]: what the first reward is formed against
    if (a.cgm0) at(a.cgm0, i) = c[0];
    store_env(a, i, pid, e);
    T l, h, r;
    risk_index1<0>(bg0, l, h, r);
    at(a.cgm, i) = c[1]; at(a.bg, i) = bg0; at(a.reward, i) = T(0); at(a.done, i) = 0;
    if (a.lbgi) at(a.lbgi, i) = l;
    if (a.hbgi) at(a.hbgi, i) = h;
    if (a.risk) at(a.risk, i) = r;
    if (a.meal) at(a.meal, i) = T(0);
    if (a.insulin) at(a.insulin, i) = T(0);
}

// T1DPatient.model at n independent points (t1d_model_rhs): the RHS of the step kernels, on its own
template <int MATH, typename T>
__global__ __launch_bounds__(kBlock) void rhs_kernel(int64_t n, const T* x, const int32_t* pid, const T* cho, const T* ins,
                                                     const T* lq, const T* lf, T* out, const T* dpar, int np, int* status)
{
    __shared__ T lds[DP_RK4_COUNT * kMaxPatients];
    for (int j = threadIdx.x; j < DP_RK4_COUNT * kMaxPatients; j += blockDim.x) lds[j] = dpar[j];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int row = pid[i];
    if ((unsigned)row >= (unsigned)np) { atomicOr(status, T1D_ST_BAD_INDEX); row = 0; }      // never index the table out of range
    ParsLds<T> p{lds, row};
    T xs[13], k[13];
#pragma unroll
    for (int j = 0; j < 13; ++j) xs[j] = x[j * n + i];
    // the per-minute inputs exactly as eat_minute forms them (t1dpatient.py:121-122,130,136-140)
    MinuteIn<T> u;
    u.d_mg = cho[i] * T(1000);
    u.ins = ins[i] * p(DP_INSC);
    const T dbar = lq[i] + lf[i] * T(1000);
    u.has_dbar = dbar > T(0);
    const T dsafe = u.has_dbar ? dbar : T(1);
    if (MATH == 0) {
        u.aa = u.has_dbar ? p(DP_CAA) / dsafe : T(0);
        u.cc = u.has_dbar ? p(DP_CCC) / dsafe : T(0);
    } else {
        const T inv2 = u.has_dbar ? fdiv(T(2), dsafe) : T(0);
        u.aa = p(DP_CAA) * inv2;
        u.cc = p(DP_CCC) * inv2;
    }
    u.bD = p(DP_B) * dsafe; u.dD = p(DP_D) * dsafe;
    u.aabD = u.aa * u.bD; u.ccdD = u.cc * u.dD;
    rhs<MATH>(p, u, xs, k);
#pragma unroll
    for (int j = 0; j < 13; ++j) out[j * n + i] = k[j];
}

__global__ void philox_normals_kernel(uint64_t seed, int64_t env_offset, int64_t n, uint32_t episode,
                                      int draw0, int n_draws, double* out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t gid = (uint64_t)(env_offset + i);
    for (int r = 0; r < n_draws; ++r) {
        const int d = draw0 + r;
        uint32_t pair; int el;
        if (d < 0) { const int q = d + 3; pair = 1u + (uint32_t)(q / 2); el = q & 1; }     // random_init_bg normals
        else if (d == 0) { pair = 0u; el = 0; }
        else { const int k = (d - 1) % 10, b = (d - 1) / 10; pair = 3u + 5u * (uint32_t)b + (uint32_t)(k / 2); el = k & 1; }
        const double2 v = philox_pair(seed, gid, episode, pair);
        out[(int64_t)r * n + i] = el ? v.y : v.x;
    }
}


// ---- RandomScenario.create_scenario for a whole batch (simulation/scenario_gen.py:33-60) ----------------
// One lane = one env: for each calendar day the episode touches, six candidate meals with presence
// probabilities (.95,.3,.95,.3,.95,.3), truncated-normal times of day (inverse-CDF sampling between the
// window bounds, rounded to the minute) and max(round(N(mu, sigma)), 0) grams.  The windows are contiguous
// and increasing, so a day's meals come out in time order; a meal landing on the minute of the one before it
// (window boundary) is dropped, as the reference's dict-by-time scenario keeps one entry per minute.
// Statistical counterpart of the reference (numpy's MT19937 stream is not reproduced); Philox subsequence =
// global env id, two blocks per candidate meal, in a key domain of its own (seed ^ kScenarioKey).
struct MealSlots {
    double prob[6], lb[6], ub[6], mu[6], sd[6], amu[6], asd[6], cdf_a[6], cdf_w[6];
};
constexpr uint64_t kScenarioKey = 0x5ce9a7105ce9a710ull;

template <typename T>
__global__ __launch_bounds__(kBlock) void random_meals_kernel(uint64_t seed, int64_t env_offset, int64_t n, int days,
                                                              const int32_t* start_tab, int start_scalar,
                                                              int32_t* meal_time, T* meal_amt, MealSlots ms)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int rows = 6 * (days + 1);
    const int start = start_tab ? start_tab[i] : start_scalar;
    const uint64_t gid = (uint64_t)(env_offset + i);
    int w = 0, last = -1;
    for (int day = 0; day <= days; ++day) {
#pragma unroll 1
        for (int k = 0; k < 6; ++k) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(seed ^ kScenarioKey, gid, 8ull * (unsigned long long)(day * 6 + k), &st);
            const double2 u = rocrand_uniform_double2(&st);           // (0, 1]
            const double2 g = rocrand_normal_double2(&st);
            const bool present = u.x <= ms.prob[k];                    // rand() < p            (:48)
            double q = ms.cdf_a[k] + u.y * ms.cdf_w[k];               // truncnorm.rvs          (:50-54)
            q = fmin(fmax(2.0 * q - 1.0, -1.0 + 1e-15), 1.0 - 1e-15);
            double tod = rint(ms.mu[k] + ms.sd[k] * 1.4142135623730951 * erfinv(q));   // np.round (:49)
            tod = fmin(fmax(tod, ms.lb[k]), ms.ub[k]);
            const double grams = fmax(rint(ms.amu[k] + ms.asd[k] * g.x), 0.0);          // :56-57
            const int minute = day * 1440 + (int)tod - start;
            if (present && minute >= 0 && minute < days * 1440 && minute != last) {
                meal_time[(int64_t)w * n + i] = minute;
                meal_amt[(int64_t)w * n + i] = (T)grams;
                last = minute; ++w;
            }
        }
    }
    for (; w < rows; ++w) { meal_time[(int64_t)w * n + i] = INT_MAX; meal_amt[(int64_t)w * n + i] = T(0); }
}

// ---- outcome statistics of a BG history on the device (analysis/report.py) ------------------------------
// One lane = one env, rows are read coalesced.  The two percentiles are exact: the order statistics are found
// by radix selection on the order-preserving integer image of the values (one pass over the env's column per
// bit), then interpolated as numpy's default 'linear' method does.
template <typename T> struct OKey;
template <> struct OKey<double> {
    typedef uint64_t U; static constexpr int bits = 64;
    static __device__ __forceinline__ U key(double v) { const U u = (U)__double_as_longlong(v); return (u >> 63) ? ~u : (u | 0x8000000000000000ull); }
    static __device__ __forceinline__ double val(U k) { const U u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k; return __longlong_as_double((long long)u); }
};
template <> struct OKey<float> {
    typedef uint32_t U; static constexpr int bits = 32;
    static __device__ __forceinline__ U key(float v) { const U u = __float_as_uint(v); return (u >> 31) ? ~u : (u | 0x80000000u); }
    static __device__ __forceinline__ float val(U k) { const U u = (k >> 31) ? (k & 0x7fffffffu) : ~k; return __uint_as_float(u); }
};

// k-th smallest (0-based) of column i, and the next order statistic after it
template <typename T>
__device__ void order_stat_pair(const T* __restrict__ tr, int64_t n, int64_t rows, int64_t i, int64_t k, T& vk, T& vk1)
{
    typedef typename OKey<T>::U U;
    U prefix = 0;
    for (int bit = OKey<T>::bits - 1; bit >= 0; --bit) {
        const U test = prefix | ((U)1 << bit);
        int64_t c = 0;
        for (int64_t r = 0; r < rows; ++r) c += OKey<T>::key(tr[r * n + i]) < test;
        if (c <= k) prefix = test;
    }
    int64_t le = 0; U next = ~(U)0; bool have = false;
    for (int64_t r = 0; r < rows; ++r) {
        const U key = OKey<T>::key(tr[r * n + i]);
        le += key <= prefix;
        if (key > prefix && (!have || key < next)) { next = key; have = true; }
    }
    vk = OKey<T>::val(prefix);
    vk1 = (le > k + 1 || !have) ? vk : OKey<T>::val(next);      // duplicates of the k-th value cover rank k + 1
}

template <typename T>
__device__ __forceinline__ T percentile_linear(const T* tr, int64_t n, int64_t rows, int64_t i, double q)
{
    const double pos = (double)(rows - 1) * q / 100.0;
    int64_t lo = (int64_t)floor(pos);
    lo = lo < 0 ? 0 : (lo > rows - 1 ? rows - 1 : lo);
    const double t = pos - (double)lo;
    T a, b;
    order_stat_pair(tr, n, rows, i, lo, a, b);
    if (lo >= rows - 1) b = a;
    const double d = (double)b - (double)a;                       // numpy _lerp
    double r = (double)a + d * t;
    if (t >= 0.5) r = (double)b - d * (1.0 - t);
    if (b == a) r = (double)a;
    return (T)r;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void outcome_kernel(int64_t n, int64_t rows, const T* __restrict__ tr, int32_t* counts,
                                                         T* pct, uint8_t* zone, T* risk_trace, double q_lo, double q_hi, int chunk)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (counts || risk_trace) {
        int c180 = 0, c70 = 0, cin = 0, c250 = 0, c50 = 0;
        double fsum = 0.0; int fcnt = 0; int64_t ch = 0;
        for (int64_t r = 0; r < rows; ++r) {
            const T bg = tr[r * n + i];
            c180 += bg > T(180); c70 += bg < T(70); cin += (bg >= T(70)) & (bg <= T(180)); c250 += bg > T(250); c50 += bg < T(50);
            if (risk_trace) {
                if (bg > T(0)) {                                                           // report.py:98-100: BG[BG > 0], and
                    const double fv = 1.509 * (pow(log((double)bg), 1.084) - 5.381);      // pandas' mean skips the NaN that
                    if (fv == fv) { fsum += fv; ++fcnt; }                                  // 0 < BG < 1 gives (negative log ^ 1.084)
                }
                if ((r + 1) % chunk == 0 || r == rows - 1) {
                    const double f = fcnt ? fsum / (double)fcnt : __builtin_nan("");
                    const double fl = f < 0.0 ? f : 0.0, fh = f > 0.0 ? f : 0.0;                        // report.py:104-105
                    risk_trace[(ch * 2) * n + i] = (T)(f == f ? 10.0 * fl * fl : f);
                    risk_trace[(ch * 2 + 1) * n + i] = (T)(f == f ? 10.0 * fh * fh : f);
                    fsum = 0.0; fcnt = 0; ++ch;
                }
            }
        }
        if (counts) { counts[i] = c180; counts[n + i] = c70; counts[2 * n + i] = cin; counts[3 * n + i] = c250; counts[4 * n + i] = c50; }
    }
    if (pct || zone) {
        const T plo = percentile_linear(tr, n, rows, i, q_lo), phi = percentile_linear(tr, n, rows, i, q_hi);
        if (pct) { pct[i] = plo; pct[n + i] = phi; }
        if (zone) {                                                     // CVGA_analysis (report.py:198-217)
            double mn = (double)plo, mx = (double)phi;
            mn = mn < 50.0 ? 50.0 : (mn > 400.0 ? 400.0 : mn);
            mx = mx < 50.0 ? 50.0 : (mx > 400.0 ? 400.0 : mx);
            const bool A = mn > 90 && mn <= 110 && mx >= 110 && mx < 180;
            const bool B = mn > 70 && mn <= 110 && mx >= 110 && mx < 300;
            const bool Cz = (mn > 90 && mn <= 110 && mx >= 300) || (mn <= 70 && mx >= 110 && mx < 180);
            const bool D = (mn > 70 && mn <= 90 && mx >= 300) || (mn <= 70 && mx >= 180 && mx < 300);
            const bool E = mn <= 70 && mx >= 300;
            zone[i] = A ? 0 : (B ? 1 : (Cz ? 2 : (D ? 3 : (E ? 4 : 5))));
        }
    }
}

} // namespace t1d
