// t1d_abi.hip -- kernels and C ABI of libt1d_hip.so (gfx950 only; see include/t1d.h).
//
// Kernels
//   step1_kernel       the headline launch: ONE simulated minute per env.step, split integrator, one persistent
//                      workgroup per CU whose waves draw 64-env chunks from a queue in LDS.
//   step_kernel        one launch per env.step, any minutes / layout / integrator: pump -> [meal bookkeeping ->
//                      n_sub sub-steps -> Gsub -> CGM sample/hold] x minutes -> risk/reward/done. (env.py:48-117)
//   refill_kernel      rebuilds due 150-minute CGM noise blocks ahead of a step kernel compiled without that code.
//   step_pipe_kernel   step_kernel made persistent with LDS-DMA prefetch (classical RK4; experiment, off by default).
//   rollout_pid_kernel n_steps x (PID or basal-bolus policy + step) with state in registers.
//   reset_kernel       masked T1DSimEnv.reset().                                              (env.py:119-155)
//   random_meals_kernel  RandomScenario.create_scenario for the whole batch.          (scenario_gen.py:33-60)
//   philox_normals_kernel  replays the Philox stream for tests.
#include "../../include/t1d.h"
#include "t1d_device.hpp"

#ifndef T1D_WAVES
#define T1D_WAVES 2
#endif
#ifndef T1D_ROW_RECOMPUTE
#define T1D_ROW_RECOMPUTE 1
#endif

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace t1d {

template <typename T> struct KArgs {
    int64_t n, env_offset;
    uint64_t seed;
    T* x; T* planned; T* last_qsto; T* last_food; int32_t* t; uint32_t* meta; uint32_t* episode; int32_t* next_meal;
    T* last_cgm; T* ar_e; T* pts; T* prev_cgm;
    const T* basal; const T* bolus; const T* cho; const int32_t* meal_time; const T* meal_amt;
    const T* normals; const T* x0_override;
    T* cgm; T* bg; T* reward; uint8_t* done; T* lbgi; T* hbgi; T* risk; T* meal; T* insulin;
    const T* dpar;          // [DP_COUNT][kMaxPatients] derived patient constants
    const T* prop;          // [prop_rows][np_pad] insulin propagator of the split integrator (kPropRows(n_sub) rows)
    const double* x0tab;    // [13][np]
    const T* minv;          // [11][11] knot second derivatives of the noise spline: M = minv . y
    int* status;
    long long* trace;       // T1D_S1_TRACE builds only: phase timestamps of the first blocks' waves
    SensorC<T> sen; PumpC<T> pump;
    int np, S, n_meals, n_normals, minutes, n_sub, flags, stagger, prop_rows, np_pad;
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

// env state held in registers across the minutes of a launch
template <typename T> struct Env {
    T x[13];
    T planned, lq, lf, last_cgm, prev_cgm;
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
    e.last_cgm = at(a.last_cgm, i); e.prev_cgm = at(a.prev_cgm, i);
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
    at(a.last_cgm, i) = e.last_cgm; at(a.prev_cgm, i) = e.prev_cgm;
    at(a.t, i) = e.t;
    if (a.next_meal && e.next_meal != e.next_meal_loaded) at(a.next_meal, i) = e.next_meal;
    at(a.meta, i) = pid | (e.eating ? T1D_META_EATING : 0u) | ((uint32_t)e.cursor << 16);
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

template <bool REFILL, typename T>
__device__ __forceinline__ T noise_sample(const KArgs<T>& a, unsigned i, int s, T (&cur)[4])
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
#pragma unroll
        for (int k = 0; k < 4; ++k) at(row(a.pts, n, 22 + k), i) = cur[k];
    }
    const T B = T(tau - 15 * m) * T(1.0 / 15.0), A = T(1) - B;
    return A * cur[0] + B * cur[1] + T(37.5) * ((A * A * A - A) * cur[2] + (B * B * B - B) * cur[3]);
}

// CGMSensor.measure (cgm.py:26-36) split in two so that the memory latency of the noise block hides
// under the ODE integration: the noise of the sample due at minute t+1 does not depend on the
// patient state, so it is drawn BEFORE the RK4 sub-steps (sample index = 1 + (t+1)/st: reset used
// #0 and #1) and added to Gsub after them.
template <bool REFILL, typename T>
__device__ __forceinline__ T measure_noise(const KArgs<T>& a, unsigned i, Env<T>& e, bool& due)
{
    const int t1 = e.t + 1;
    int q, r;
    divmod_uniform(t1, a.sen.st, q, r);
    due = r == 0;
    return due ? noise_sample<REFILL>(a, i, 1 + q, e.cur) : T(0);
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
template <typename T>
__device__ __forceinline__ T meal_lookup(const KArgs<T>& a, unsigned i, Env<T>& e)
{
    T meal = T(0);
    if (a.next_meal) {
        if (e.next_meal <= e.t) {                       // rare: a meal fires (or stale entries are skipped)
            while (e.cursor < a.n_meals) {
                const int mt = at(rowv(a.meal_time, a.n, e.cursor), i);
                if (mt > e.t) { e.next_meal = mt; break; }
                if (mt == e.t) meal = at(rowv(a.meal_amt, a.n, e.cursor), i);
                ++e.cursor;
            }
            if (e.cursor >= a.n_meals) e.next_meal = INT_MAX;
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
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `pre_rk4` runs once, immediately before the first minute's RK4 sub-steps: from there to the end of
// the integration the wave issues no vector-memory instruction, which is where the persistent kernel
// starts the LDS-DMA of its next tile.
template <int MATH, typename T, typename P, typename Hook = NoHook, bool LOCALP = false, bool REFILL = true, typename PR = NoProp>
__device__ __forceinline__ StepOut<T> step_body(const KArgs<T>& a, P& p, unsigned i, Env<T>& e,
                                                T basal, T bolus, bool has_bolus, Hook pre_rk4 = Hook(), PR pr = PR())
{
    T q_basal, q_bolus;
    if (a.flags & (T1D_BATCH_NO_PUMP | 0x200)) { // T1DPatient.step driven directly: insulin = basal + bolus as given
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
        const T noise = (a.flags & 0x400) ? (due = false, T(0)) : measure_noise<REFILL>(a, i, e, due);
        MinuteIn<T> u = eat_minute<MATH, T>(p, e.x, meal, insulin, e.planned, e.lq, e.lf, e.eating);
        // every load this minute issued is needed by the integration anyway: drain them HERE, on every
        // path, so that the compiler's own wait cannot land behind the hook's (invisible) DMA instructions
        if constexpr (PR::kSplit) p.pin_split(); else p.pin();
        {   // volatile asms keep their order: everything the RK4 loop consumes is computed (and any spilled
            // operand reloaded) BEFORE the hook below issues its DMA
            T aa = u.aa, cc = u.cc, bD = u.bD, dD = u.dD, dmg = u.d_mg, ins = u.ins;
            asm volatile("" : "+v"(aa), "+v"(cc), "+v"(bD), "+v"(dD), "+v"(dmg), "+v"(ins));
            u.aa = aa; u.cc = cc; u.bD = bD; u.dD = dD; u.d_mg = dmg; u.ins = ins;
        }
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
        if (m == 0) pre_rk4();
        if constexpr (PR::kSplit) { if (!(a.flags & 0x800)) split_minute(p, pr, u, e.x, a.n_sub); }
        else { if (!(a.flags & 0x800)) rk4_minute<MATH>(p, u, e.x, a.n_sub, LOCALP); }
        e.t += 1;
        const T gsub = MATH == 0 ? e.x[12] / p(DP_VG) : e.x[12] * p(DP_IVG);      // t1dpatient.py:217-218
        const T cgm = measure_apply(a, e, gsub, noise, due);                      // env.py:62
        if (MATH == 0) { o.meal += meal / div; o.ins += insulin / div; o.bg += gsub / div; o.cgm += cgm / div; }   // env.py:78-81
        else { o.meal += meal * inv_div; o.ins += insulin * inv_div; o.bg += gsub * inv_div; o.cgm += cgm * inv_div; }
    }
    if (MATH != 0 && a.minutes == 1) { o.ins = insulin; }   // x * (1/1) is exact already; keeps -0 out
    return o;
}

// risk of CGM_hist[-1] (risk_diff, env.py:27-33): independent of this step's integration, so callers
// evaluate it BEFORE the minute loop, where it overlaps with the pump / meal / noise chains
template <int MATH, typename T>
__device__ __forceinline__ T prev_risk(const KArgs<T>& a, T prev_cgm)
{
    T l, h, rp = T(0);
    if (!(a.flags & 0x100)) risk_index1<MATH>(prev_cgm, l, h, rp);
    return rp;
}

template <int MATH, typename T>
__device__ __forceinline__ void write_outputs(const KArgs<T>& a, unsigned i, Env<T>& e, const StepOut<T>& o, T rp)
{
    T l, h, r, rc = T(0);
    if (!(a.flags & 0x100)) risk_index1<MATH>(o.cgm, l, h, rc);
    at(a.reward, i) = rp - rc;
    e.prev_cgm = o.cgm;
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

// VARIANT 0: reference arithmetic (ocml tanh, IEEE divisions), parameters from LDS
//         1: fast arithmetic, parameters re-read from LDS per RHS evaluation (any patient layout)
//         2: fast arithmetic, wave-uniform patient, parameters in SGPRs (T1D_BATCH_WAVE_UNIFORM)
//         3: fast arithmetic, parameters gathered once per lane into VGPRs (any patient layout)
//         4: as 3, split integrator (t1d_device.hpp) with the insulin propagator staged in LDS
//         5: as 1, split integrator
//         6, 7: as 4, 5 with the adaptive gut refinement
template <int VARIANT> struct VariantMath { static constexpr int value = VARIANT == 0 ? 0 : 1; };
template <int VARIANT> struct VariantInfo {
    static constexpr bool split = VARIANT >= 4 && VARIANT <= 7;
    static constexpr bool adapt = VARIANT == 6 || VARIANT == 7;
    static constexpr bool lds_pars = VARIANT == 0 || VARIANT == 1 || VARIANT == 5 || VARIANT == 7;
    static constexpr bool reg_pars = VARIANT == 3 || VARIANT == 4 || VARIANT == 6;
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
    const T rp = prev_risk<MATH>(a, e.prev_cgm);
    StepOut<T> o;
    if constexpr (VARIANT == 4 || VARIANT == 6) {
        ParsReg<T> p;
        p.load(a.dpar, (int)pid);
        PropLds<T, VI::adapt> pr{(const T*)t1d_dyn_lds, a.np_pad, (int)pid};
        o = step_body<MATH, T, ParsReg<T>, NoHook, false, REFILL, PropLds<T, VI::adapt>>(a, p, i, e, basal, bolus, a.bolus != nullptr, NoHook(), pr);
    } else if constexpr (VARIANT == 5 || VARIANT == 7) {
        ParsLds<T> p{lds, (int)pid};
        PropLds<T, VI::adapt> pr{(const T*)t1d_dyn_lds, a.np_pad, (int)pid};
        o = step_body<MATH, T, ParsLds<T>, NoHook, false, REFILL, PropLds<T, VI::adapt>>(a, p, i, e, basal, bolus, a.bolus != nullptr, NoHook(), pr);
    } else if constexpr (VARIANT == 2) {
        const int pid0 = __builtin_amdgcn_readfirstlane((int)pid);
        if (__ballot((int)pid != pid0) != 0ull) { atomicOr(a.status, T1D_ST_BAD_LAYOUT); return; }
        ParsScalar<T> p;
        p.load(a.dpar, kMaxPatients, pid0);
        o = step_body<MATH, T, ParsScalar<T>, NoHook, false, REFILL>(a, p, i, e, basal, bolus, a.bolus != nullptr);
    } else if constexpr (VARIANT == 3) {
        ParsReg<T> p;
        p.load(a.dpar, (int)pid);
        o = step_body<MATH, T, ParsReg<T>, NoHook, false, REFILL>(a, p, i, e, basal, bolus, a.bolus != nullptr);
    } else {
        ParsLds<T> p{lds, (int)pid};
        o = step_body<MATH, T, ParsLds<T>, NoHook, false, REFILL>(a, p, i, e, basal, bolus, a.bolus != nullptr);
    }
    write_outputs<MATH>(a, i, e, o, rp);
    store_env(a, i, pid, e);
}

// ---- single-minute step, split integrator, persistent blocks -----------------------------------------
// The launch the headline workload makes a million times: one simulated minute per env.step (1-min sensors),
// no noise-block refill due (refill_kernel ran, or the host vouched).  Differences from step_kernel:
//   * blocks are persistent (grid = what is resident) and walk tiles of 256 envs, so the parameter and
//     propagator tables are staged into LDS once per block instead of once per 256 envs, compactly
//     (row stride 32 or 64 patients, a compile-time constant: every table read is a ds_read with an
//     immediate offset);
//   * everything the integration does not need is stored BEFORE it (meal bookkeeping, clock, meal cursor,
//     insulin/meal outputs), so that only the 13 states, the drawn noise and the previous risk are alive
//     across the sub-step loops -- which is what lets four waves share a SIMD (<= 128 VGPRs) where
//     step_kernel needs ~230.
#ifndef T1D_S1_WAVES
#define T1D_S1_WAVES 3
#endif
#ifndef T1D_S1_TRACE
#define T1D_S1_TRACE 0
#endif
#ifndef T1D_S1_ROTATE_PRIO
#define T1D_S1_ROTATE_PRIO 1
#endif
#if T1D_S1_TRACE
// tuning builds: drain every counter and stamp the wall clock (100 MHz) at phase boundaries
#define S1_MARK(m) do { __builtin_amdgcn_s_waitcnt(0x0070); if (tr && (threadIdx.x & 63) == 0 && tk < 8) tr[tk * 8 + (m)] = (long long)wall_clock64(); } while (0)
#else
#define S1_MARK(m) do { } while (0)
#endif
constexpr int kS1Threads = 256 * T1D_S1_WAVES;        // one workgroup fills a CU: T1D_S1_WAVES waves on each of its 4 SIMDs
// EXTRA: the optional outputs (lbgi, hbgi, risk, meal, insulin) exist; without them their five pointers and the
// third risk evaluation drop out of the kernel altogether
template <bool REG, typename T, int STRIDE, bool EXTRA, bool ADAPT>
__global__ __launch_bounds__(kS1Threads, 1) void step1_kernel(const KArgs<T> a, int nchunks)
{
    // packed state only (t1d_step checks): rows 13.. of the x buffer are planned, last_qsto, last_food, last_cgm,
    // prev_cgm and the 26 noise rows; rows 1, 2 of the t buffer are meta and next_meal.  Deriving them from two
    // base pointers instead of reading ten more kernel arguments keeps the scalar registers from spilling.
    T* const ldp = (T*)t1d_dyn_lds;                        // [DP_COUNT][STRIDE]
    T* const lpr = ldp + DP_COUNT * STRIDE;                // [prop_rows][STRIDE]
    __shared__ int queue;
    __shared__ T lconst[8];                              // pump and sensor limits: read from LDS where used, so that they
                                                         // do not sit in (spilled) scalar registers across the whole kernel
    for (int j = threadIdx.x; j < DP_COUNT * STRIDE; j += kS1Threads) {
        const int r = j / STRIDE, c = j % STRIDE;
        ldp[j] = c < a.np ? a.dpar[r * kMaxPatients + c] : T(0);
    }
    for (int j = threadIdx.x; j < a.prop_rows * STRIDE; j += kS1Threads) {
        const int r = j / STRIDE, c = j % STRIDE;
        lpr[j] = c < a.np ? a.prop[r * a.np_pad + c] : T(0);
    }
    if (threadIdx.x == 0) {
        queue = 0;
        lconst[0] = a.pump.inc_basal; lconst[1] = a.pump.min_basal; lconst[2] = a.pump.max_basal;
        lconst[3] = a.pump.inc_bolus; lconst[4] = a.pump.min_bolus; lconst[5] = a.pump.max_bolus;
        lconst[6] = a.sen.vmin; lconst[7] = a.sen.vmax;
    }
    __syncthreads();
    // This workgroup owns a contiguous run of 64-env chunks; its waves draw them from a queue in LDS.  The
    // SIMD issues oldest-wave-first, so with a fixed share per wave the first wave of a SIMD would race ahead
    // and the last would finish alone (measured: 66 vs 93 us); with the queue the fast wave simply takes more.
    const int per_block = (nchunks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int first = (int)blockIdx.x * per_block;
    const int count = nchunks - first < per_block ? nchunks - first : per_block;
    const unsigned lane = threadIdx.x & 63u;
#if T1D_S1_TRACE
    long long* tr = (a.trace && blockIdx.x < 32) ? a.trace + (blockIdx.x * (kS1Threads / 64) + threadIdx.x / 64) * 64 : nullptr;
    int tk = -1;
#endif
    for (int it = 0;; ++it) {
        int c = 0;
        if (lane == 0) c = atomicAdd(&queue, 1);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= count) break;                              // wave-uniform
        const unsigned i = (unsigned)(first + c) * 64u + lane;
        __builtin_assume(i < (1u << 28));
        if ((int64_t)i >= a.n) continue;
#if T1D_S1_TRACE
        ++tk;
#endif
#if T1D_S1_ROTATE_PRIO
        // rotate the issue priority among the waves of a SIMD (waves w, w + 4, w + 8 of the workgroup) chunk by chunk
        switch ((unsigned)(it + (int)(threadIdx.x >> 8)) % 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(2); break;
        }
#endif
        S1_MARK(0);
        const uint32_t meta = at(row(a.t, a.n, 1), i);
        const uint32_t pid = T1D_META_PID(meta);
        Env<T> e;
#pragma unroll
        for (int k = 0; k < 13; ++k) e.x[k] = at(row(a.x, a.n, k), i);
        e.planned = at(row(a.x, a.n, 13), i); e.lq = at(row(a.x, a.n, 14), i); e.lf = at(row(a.x, a.n, 15), i);
        const T planned0 = e.planned, lq0 = e.lq, lf0 = e.lf;
        e.t = at(a.t, i);
        e.next_meal = at(row(a.t, a.n, 2), i);
        e.next_meal_loaded = e.next_meal;
        e.eating = (meta & T1D_META_EATING) != 0;
        e.cursor = (int)T1D_META_CURSOR(meta);
        const T basal = at(a.basal, i);
        const T bolus = a.bolus ? at(a.bolus, i) : T(0);
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
        const T meal = a.cho ? at(a.cho, i) : meal_lookup(a, i, e);                               // env.py:50
        ParsLdsS<T, STRIDE> pl{ldp, (int)pid};
        MinuteIn<T> u = eat_minute<1, T>(pl, e.x, meal, insulin, e.planned, e.lq, e.lf, e.eating);
        // bookkeeping is final for this minute: store it now -- the meal words only where they changed (they do
        // while an env is eating, ~3 % of the minutes: 24 B per env-step of write traffic otherwise)
        if (e.planned != planned0) at(row(a.x, a.n, 13), i) = e.planned;
        if (e.lq != lq0) at(row(a.x, a.n, 14), i) = e.lq;
        if (e.lf != lf0) at(row(a.x, a.n, 15), i) = e.lf;
        at(a.t, i) = e.t + 1;
        if (e.next_meal != e.next_meal_loaded) at(row(a.t, a.n, 2), i) = e.next_meal;
        {   // patient id, eating flag, meal cursor: changes when a meal starts, ends or fires
            const uint32_t meta1 = pid | (e.eating ? T1D_META_EATING : 0u) | ((uint32_t)e.cursor << 16);
            if (meta1 != meta) at(row(a.t, a.n, 1), i) = meta1;
        }
        if (EXTRA) {
            if (a.meal) at(a.meal, i) = meal;
            if (a.insulin) at(a.insulin, i) = insulin;
        }
        S1_MARK(2);
        {
            PropLdsS<T, STRIDE, ADAPT> pr{lpr, (int)pid};
            if (REG) {
                ParsReg<T> p;
#pragma unroll
                for (int k = 0; k < (int)(sizeof(kSplitPars) / sizeof(int)); ++k) p.v[kSplitPars[k]] = pl(kSplitPars[k]);
                if (ADAPT) {
#pragma unroll
                    for (int k = 0; k < (int)(sizeof(kAdaptPars) / sizeof(int)); ++k) p.v[kAdaptPars[k]] = pl(kAdaptPars[k]);
                }
                p.pin_split();
                if (!(a.flags & 0x800)) split_minute(p, pr, u, e.x, a.n_sub);
            } else {
                if (!(a.flags & 0x800)) split_minute(pl, pr, u, e.x, a.n_sub);
            }
        }
        S1_MARK(3);
#pragma unroll
        for (int k = 0; k < 13; ++k) at(row(a.x, a.n, k), i) = e.x[k];
        // the sensor side is fetched only now: nothing of it has to stay in registers across the integration
#pragma unroll
        for (int k = 0; k < 4; ++k) e.cur[k] = at(row(a.x, a.n, 40 + k), i);
        // with a 1-minute sensor every minute takes a fresh sample: the held value is never read
        T last_cgm = a.sen.st == 1 ? T(0) : (T)at(row(a.x, a.n, 16), i);
        const T prev_cgm = at(row(a.x, a.n, 17), i);
        S1_MARK(4);
        bool due;
        const T noise = measure_noise<false>(a, i, e, due);       // e.t is still the minute's start: sample for t + 1
        const T rp = prev_risk<1>(a, prev_cgm);
        const T gsub = e.x[12] * pl(DP_IVG);                                                       // t1dpatient.py:217-218
        if (due) {                                                                                 // cgm.py:26-36
            T c = gsub + noise;
            int z = 0;
            asm volatile("" : "+v"(z));
            const T vmin = lconst[6 + z], vmax = lconst[7 + z];
            c = c > vmin ? c : vmin;
            c = c < vmax ? c : vmax;
            last_cgm = c;
            if (a.sen.st != 1) at(row(a.x, a.n, 16), i) = c;      // the zero-order hold is dead state with a 1-minute sensor
        }
        T l, h, r, rc = T(0);
        if (!(a.flags & 0x100)) risk_index1<1>(last_cgm, l, h, rc);
        at(a.reward, i) = rp - rc;                                                                 // env.py:27-33
        at(row(a.x, a.n, 17), i) = last_cgm;
        at(a.cgm, i) = last_cgm; at(a.bg, i) = gsub;
        at(a.done, i) = (gsub < T(70) || gsub > T(350)) ? 1 : 0;                                   // env.py:103
        if (EXTRA && (a.lbgi || a.hbgi || a.risk)) {
            risk_index1<1>(gsub, l, h, r);                                                         // env.py:85
            if (a.lbgi) at(a.lbgi, i) = l;
            if (a.hbgi) at(a.hbgi, i) = h;
            if (a.risk) at(a.risk, i) = r;
        }
        if (!(fabs((double)e.x[12]) <= 1.0e300)) atomicOr(a.status, T1D_ST_NONFINITE);
#if T1D_S1_TRACE
        if (tr && (threadIdx.x & 63) == 0 && tk < 8) tr[tk * 8 + 5] = (long long)wall_clock64();    // epilogue computed, stores issued
#endif
        S1_MARK(6);
    }
}

// ---- persistent, software-pipelined step ---------------------------------------------------------
// One-tile-per-block launches keep the two waves of a SIMD in lock-step: both wait for their loads,
// then both compete for the VALU, then both store, so the chip alternates between an idle VALU and an
// idle memory system (measured: ~60-90 us of a 150-170 us launch at 1 Mi envs).  Here each block walks
// tiles blockIdx.x, +gridDim.x, ... and every wave streams the state of its NEXT 64 envs from HBM
// straight into a wave-private LDS staging area with LDS-DMA (global_load_lds_dwordx4: no VGPR is
// held by data in flight) while it integrates the current 64.  Nothing but the issuing wave's vmcnt
// orders a ds_read behind a pending LDS-DMA, hence the explicit waits.
typedef __attribute__((address_space(1))) const void t1d_gptr;
typedef __attribute__((address_space(3))) void t1d_lptr;

// The pipelined kernel needs the per-env state PACKED: one [44][n] buffer of T (rows 0-12 x, 13 planned,
// 14 last_qsto, 15 last_food, 16 last_cgm, 17 prev_cgm, 18-43 pts) and one [3+][n] int buffer (t, meta,
// next_meal), so that every staged row is `base + 32-bit offset` (t1d_step checks the pointers and
// falls back to step_kernel otherwise).  Stage rows: 0-17 = state rows 0-17, 18-21 = pts rows 22-25
// (state rows 40-43), 22.. = basal, 22+G.. = bolus (each DMA group fetches G rows: 2 for double, 4 for float).
constexpr int kPackedRows = 18 + kPtsRows;            // 44
template <typename T> struct StageGeom {
    static constexpr int EPL = 16 / (int)sizeof(T);   // elements per lane per DMA
    static constexpr int LPR = 64 / EPL;              // lanes per 64-element row
    static constexpr int G = 64 / LPR;                // rows per DMA instruction (2 / 4)
    static constexpr int BASAL = 24;                  // first stage row of the basal group
    static constexpr int BOLUS = BASAL + G;
    static constexpr int ROWS = BOLUS + G;
};
template <typename T> struct Stage {
    T f[StageGeom<T>::ROWS][64];
    int i[4][64];
};

// One LDS-DMA: lane l fetches 16 B at base + voff; the 1 KiB lands contiguously at lds_dst (M0).
// Issued through inline asm on purpose: when hipcc knows about a pending LDS-DMA it puts an
// `s_waitcnt vmcnt(0)` in front of EVERY later LDS read (here: the parameter table inside the RK4
// loop), which serialises the prefetch with the arithmetic it is meant to hide under.  The loop in
// step_pipe_kernel counts and waits for these operations itself.
__device__ __forceinline__ void dma16(const void* base_uniform, unsigned voff, void* lds_dst)
{
    const unsigned lds_addr = (unsigned)(size_t)(t1d_lptr*)lds_dst;
    unsigned saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "v"(voff), "s"(base_uniform), "s"(lds_addr) : "memory");
}

template <typename T>
__device__ __forceinline__ void stage_tile(const KArgs<T>& a, unsigned elem0, Stage<T>* st)
{
    using GEO = StageGeom<T>;
    unsigned lane = threadIdx.x & 63u;
    asm volatile("" : "+v"(lane));       // recompute the few offsets here: hoisted out of the tile loop they get spilled,
                                         // and a scratch reload between two DMAs is a vmcnt(0) that waits for the first
    const unsigned rowb = (unsigned)a.n * (unsigned)sizeof(T);                       // bytes per state row
    const unsigned v0 = elem0 * (unsigned)sizeof(T) + (lane / GEO::LPR) * rowb + (lane % GEO::LPR) * 16u;
    const void* xb = a.x;
#pragma unroll
    for (int r = 0; r < 18; r += GEO::G) dma16(xb, v0 + (unsigned)r * rowb, &st->f[r][0]);   // rows 0..17 (+ spill-over into 18, 19 for float)
#pragma unroll
    for (int r = 0; r < 4; r += GEO::G) dma16(xb, v0 + (unsigned)(40 + r) * rowb, &st->f[20 + r][0]);
    const unsigned vsame = elem0 * (unsigned)sizeof(T) + (lane % GEO::LPR) * 16u;  // every row group = the same row
    dma16(a.basal, vsame, &st->f[GEO::BASAL][0]);
    if (a.bolus) dma16(a.bolus, vsame, &st->f[GEO::BOLUS][0]);
    const unsigned rowi = (unsigned)a.n * 4u;
    const unsigned li = lane / 16u;
    dma16(a.t, elem0 * 4u + (li < 3u ? li : 0u) * rowi + (lane % 16u) * 16u, &st->i[0][0]);
}

template <typename T>
__device__ __forceinline__ void unstage(const KArgs<T>& a, const Stage<T>* st, Env<T>& e, T& basal, T& bolus, uint32_t& meta)
{
    using GEO = StageGeom<T>;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 13; ++k) e.x[k] = st->f[k][lane];
    e.planned = st->f[13][lane]; e.lq = st->f[14][lane]; e.lf = st->f[15][lane];
    e.last_cgm = st->f[16][lane]; e.prev_cgm = st->f[17][lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) e.cur[k] = st->f[20 + k][lane];
    basal = st->f[GEO::BASAL][lane];
    bolus = a.bolus ? st->f[GEO::BOLUS][lane] : T(0);
    e.t = st->i[0][lane];
    meta = (uint32_t)st->i[1][lane];
    e.next_meal = st->i[2][lane];
    e.next_meal_loaded = e.next_meal;
    e.eating = (meta & T1D_META_EATING) != 0;
    e.cursor = (int)T1D_META_CURSOR(meta);
}

// requires a.n % kBlock == 0 and the packed state layout (the host falls back to step_kernel otherwise)
template <int VARIANT, typename T>
__global__ __launch_bounds__(kBlock, T1D_WAVES) void step_pipe_kernel(const KArgs<T> a)
{
    constexpr int MATH = 1;
    __shared__ T lds[VARIANT == 2 ? 1 : DP_RK4_COUNT * kMaxPatients];
    __shared__ Stage<T> stage[kBlock / 64];
    if (VARIANT != 2) stage_pars(a, lds, DP_RK4_COUNT);
    const unsigned ntiles = (unsigned)(a.n / kBlock);
    unsigned tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    Stage<T>* st = &stage[wave];
    // Identical waves that start together stay in lock-step: both waves of a SIMD are in their latency-bound
    // prologue/epilogue at the same time and in the VALU-dense RK4 loop at the same time.  Delaying the
    // second half of the grid (the second workgroup of each CU under round-robin dispatch) by a fraction
    // of a tile puts the pairs out of phase for the rest of the launch.
    if (a.stagger > 0 && blockIdx.x >= (gridDim.x + 1) / 2)
        for (int k = 0; k < a.stagger; ++k) __builtin_amdgcn_s_sleep(127);
    stage_tile(a, tile * kBlock + (unsigned)wave * 64u, st);
    bool first = true;
    for (;;) {
        const unsigned i = tile * kBlock + threadIdx.x;
        __builtin_assume(i < (1u << 28));
        Env<T> e;
        T basal, bolus;
        uint32_t meta;
        // This tile's DMA must have landed.  vmcnt retires in order and the DMA is OLDER than everything the
        // previous tile issued afterwards, of which at least kTileStores are unconditional stores
        // (x[13], planned, last_qsto, last_food, last_cgm, prev_cgm, t, meta, cgm, bg, reward, done): once
        // at most that many operations are outstanding the DMA is complete, and the wave does not sit
        // through the write burst of its own stores.
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        unstage(a, st, e, basal, bolus, meta);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // staging area read out before it is refilled
        __builtin_amdgcn_sched_barrier(0);
        const unsigned ntile = tile + gridDim.x;
        const bool more = ntile < ntiles;
        const unsigned next_elem0 = ntile * kBlock + (unsigned)wave * 64u;
        auto prefetch = [&]() {
            if (more) stage_tile(a, next_elem0, st);      // in flight while this tile integrates
        };
        const uint32_t pid = T1D_META_PID(meta);
        const T rp = prev_risk<MATH>(a, e.prev_cgm);
        StepOut<T> o;
        if (VARIANT == 2) {
            const int pid0 = __builtin_amdgcn_readfirstlane((int)pid);
            if (__ballot((int)pid != pid0) != 0ull) {
                atomicOr(a.status, T1D_ST_BAD_LAYOUT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                break;                                   // no stores were issued: the counted wait above would not hold
            }
            ParsScalar<T> p;
            p.load(a.dpar, kMaxPatients, pid0);
            o = step_body<MATH>(a, p, i, e, basal, bolus, a.bolus != nullptr, prefetch);
            write_outputs<MATH>(a, i, e, o, rp);
            store_env(a, i, pid, e);
        } else if (VARIANT == 3) {
            ParsReg<T> p;
            p.load(lds, (int)pid);                       // 38 ds_reads per tile, then no LDS traffic in the RK4 loop
            o = step_body<MATH>(a, p, i, e, basal, bolus, a.bolus != nullptr, prefetch);
            write_outputs<MATH>(a, i, e, o, rp);
            store_env(a, i, pid, e);
        } else {
            ParsLds<T> p{lds, (int)pid};
            o = step_body<MATH>(a, p, i, e, basal, bolus, a.bolus != nullptr, prefetch);
            write_outputs<MATH>(a, i, e, o, rp);
            store_env(a, i, pid, e);
        }
        if (!more) break;
        tile = ntile;
        first = false;
    }
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
    T pre_prev_cgm = e.prev_cgm;
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
        o = step_body<MATH, T, P, NoHook, false, true, PR>(a, p, i, e, u, bolus, true, NoHook(), pr);
        obs = o.cgm;
        prev_meal = o.meal;
        if (c.bg_trace) c.bg_trace[(c.trace_row + s) * a.n + i] = o.bg;
        if (c.cgm_trace) c.cgm_trace[(c.trace_row + s) * a.n + i] = o.cgm;
        if (c.cho_trace) c.cho_trace[(c.trace_row + s) * a.n + i] = o.meal;
        if (c.ins_trace) c.ins_trace[(c.trace_row + s) * a.n + i] = o.ins;
        pre_prev_cgm = e.prev_cgm;
        e.prev_cgm = o.cgm;                      // CGM history advances every step
        if (c.sum_risk) { T l, h, r; risk_index1<MATH>(o.bg, l, h, r); sum_risk += r; }
        min_bg = o.bg < min_bg ? o.bg : min_bg;
        max_bg = o.bg > max_bg ? o.bg : max_bg;
        n_low += o.bg < T(70); n_high += o.bg > T(180);
    }
    e.prev_cgm = pre_prev_cgm;                   // the last step's reward is formed from it
    write_outputs<MATH>(a, i, e, o, prev_risk<MATH>(a, pre_prev_cgm));
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
        rollout_body<VARIANT>(a, c, p, i, pid, e, PropLds<T, VI::adapt>{(const T*)t1d_dyn_lds, a.np_pad, (int)pid});
    } else if constexpr (VARIANT == 5 || VARIANT == 7) {
        ParsLds<T> p{lds, (int)pid};
        rollout_body<VARIANT>(a, c, p, i, pid, e, PropLds<T, VI::adapt>{(const T*)t1d_dyn_lds, a.np_pad, (int)pid});
    } else if constexpr (VARIANT == 2) {
        const int pid0 = __builtin_amdgcn_readfirstlane((int)pid);
        if (__ballot((int)pid != pid0) != 0ull) { atomicOr(a.status, T1D_ST_BAD_LAYOUT); return; }
        ParsScalar<T> p;
        p.load(a.dpar, kMaxPatients, pid0);
        rollout_body<VARIANT>(a, c, p, i, pid, e);
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
    e.prev_cgm = c[0];
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
                if (bg > T(0)) { fsum += 1.509 * (pow(log((double)bg), 1.084) - 5.381); ++fcnt; }      // report.py:98-100
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

// =============================================================================================
// host side
// =============================================================================================
using namespace t1d;

struct t1d_ctx {
    int device = -1;
    int np = 0, S = 0;
    double sensor[T1D_SENSOR_NCOLS];
    double pump[T1D_PUMP_NCOLS];
    double* d_par64 = nullptr; float* d_par32 = nullptr;
    double* d_x0 = nullptr;
    double* d_minv64 = nullptr; float* d_minv32 = nullptr;
    int* d_status = nullptr;
    int math = 1;            // RHS arithmetic variant (t1d_ctx_set_option "math")
    int scalar_params = 0;   // 1 = wave-uniform batches use the SGPR-parameter kernels
    int params_mode = -1;    // 0 = LDS re-read per RHS evaluation, 1 = gathered once into VGPRs, -1 = by minutes per launch
    int pipeline = 0;        // 1 = persistent LDS-DMA pipelined step kernel, 0 = one tile per block
    int n_cu = 256;
    int pipe_blocks = 0;     // > 0: grid of the persistent kernel (tests exercise several tiles per block)
    int split_refill = 1;    // 1 = noise-block refills run in their own kernel ahead of a refill-free step kernel
    int pipe_stagger = 0;    // s_sleep(127) iterations (~3.4 us each) by which the second half of the persistent grid starts late
    int adaptive_gut = 1;    // 1 (default) = the split integrator halves the gut step in minutes that cross a gastric-emptying transition fast
    int single_minute_kernel = 1;   // 1 = minutes == 1 launches of the split integrator use the persistent early-store kernel
    int integrator = -1;     // 0 = classical RK4 on all 13 states, 1 = split scheme, -1 = split whenever n_sub allows it
    int split_nsub = 0;      // n_sub the split tables on the device were built for (0 = none yet)
    int np_pad = 0;
    double* d_prop64 = nullptr; float* d_prop32 = nullptr;   // [kPropRows(split_nsub)][np_pad]
    long long* d_trace = nullptr;    // T1D_S1_TRACE builds
    std::vector<double> ptab;    // the caller's table, kept for rebuilding the split tables
    std::vector<double> dpar;    // host copy of the derived-parameter table
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define T1D_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t _e = (call);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(T1D_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e));        \
    } while (0)

// Minv [11][11] with M = Minv . y: second derivatives at the knots of the not-a-knot cubic spline
// through 11 points at 15-minute spacing -- what scipy's interp1d(kind='cubic') builds in
// noise_gen.py:45.  Interior rows: M[k-1] + 4 M[k] + M[k+1] = 6 (y[k-1] - 2 y[k] + y[k+1]) / h^2; the two
// not-a-knot rows make the third derivative continuous at the first and last interior knots.
static std::vector<double> spline_second_derivative_operator()
{
    const int K = 11;
    const double h = 15.0;
    std::vector<double> A(K * K, 0.0), B(K * K, 0.0);
    for (int k = 1; k < K - 1; ++k) {
        A[k * K + k - 1] = 1.0; A[k * K + k] = 4.0; A[k * K + k + 1] = 1.0;
        B[k * K + k - 1] = 6.0 / (h * h); B[k * K + k] = -12.0 / (h * h); B[k * K + k + 1] = 6.0 / (h * h);
    }
    A[0] = 1.0; A[1] = -2.0; A[2] = 1.0;
    A[(K - 1) * K + K - 3] = 1.0; A[(K - 1) * K + K - 2] = -2.0; A[(K - 1) * K + K - 1] = 1.0;
    // Gauss-Jordan with partial pivoting on [A | B]
    for (int c = 0; c < K; ++c) {
        int piv = c;
        for (int r = c + 1; r < K; ++r) if (std::fabs(A[r * K + c]) > std::fabs(A[piv * K + c])) piv = r;
        if (piv != c) for (int j = 0; j < K; ++j) { std::swap(A[c * K + j], A[piv * K + j]); std::swap(B[c * K + j], B[piv * K + j]); }
        const double d = A[c * K + c];
        for (int j = 0; j < K; ++j) { A[c * K + j] /= d; B[c * K + j] /= d; }
        for (int r = 0; r < K; ++r) {
            if (r == c) continue;
            const double f = A[r * K + c];
            if (f == 0.0) continue;
            for (int j = 0; j < K; ++j) { A[r * K + j] -= f * A[c * K + j]; B[r * K + j] -= f * B[c * K + j]; }
        }
    }
    return B;
}


// ---- host tables of the split integrator ------------------------------------------------------------
// exp(A) for a small dense matrix: scaling and squaring with a degree-16 Taylor polynomial
static void mat_expm(int n, const double* A, double* E)
{
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) { double r = 0.0; for (int j = 0; j < n; ++j) r += std::fabs(A[i * n + j]); nrm = std::max(nrm, r); }
    int sq = 0;
    while (nrm > 0.03125 && sq < 60) { nrm *= 0.5; ++sq; }
    const double sc = std::ldexp(1.0, -sq);
    std::vector<double> B(n * n), term(n * n, 0.0), tmp(n * n);
    for (int k = 0; k < n * n; ++k) B[k] = A[k] * sc;
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) E[i * n + j] = (i == j); term[i * n + i] = 1.0; }
    for (int d = 1; d <= 16; ++d) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double v = 0.0;
                for (int k = 0; k < n; ++k) v += term[i * n + k] * B[k * n + j];
                tmp[i * n + j] = v / (double)d;
            }
        term = tmp;
        for (int k = 0; k < n * n; ++k) E[k] += term[k];
    }
    for (int q = 0; q < sq; ++q) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double v = 0.0;
                for (int k = 0; k < n; ++k) v += E[i * n + k] * E[k * n + j];
                tmp[i * n + j] = v;
            }
        for (int k = 0; k < n * n; ++k) E[k] = tmp[k];
    }
}

// One patient row -> kPropRows(ng) propagator entries (layout: t1d_device.hpp) followed by the four x2
// weights E, wa, wm, wb for h = 1/ng and the same four for h/2 (adaptive gut refinement).  The insulin sub-system in the order s = (x5, x9, x10, x11, x6, x7, x8, u, 1)
// (t1dpatient.py:176-198); weights of x2' = -kabs x2 + F (:148) for h = 1/ng from the moments
// I_k = int_0^1 exp(-z (1 - s)) s^k ds = sum_j (-z)^j k! / (k + j + 1)!,  z = kabs h, of the quadratic
// through F(0), F(h/2), F(h).
static void split_tables_row(const double* r, int ng, double* out)
{
    double A[81] = {0.0};
    auto at = [&](int i, int j) -> double& { return A[i * 9 + j]; };
    at(0, 0) = -(r[T1D_P_M2] + r[T1D_P_M4]); at(0, 1) = r[T1D_P_M1]; at(0, 2) = r[T1D_P_KA1]; at(0, 3) = r[T1D_P_KA2];
    at(1, 1) = -(r[T1D_P_M1] + r[T1D_P_M30]); at(1, 0) = r[T1D_P_M2];
    at(2, 2) = -(r[T1D_P_KA1] + r[T1D_P_KD]); at(2, 7) = 1.0;
    at(3, 2) = r[T1D_P_KD]; at(3, 3) = -r[T1D_P_KA2];
    at(4, 4) = -r[T1D_P_P2U]; at(4, 0) = r[T1D_P_P2U] / r[T1D_P_VI]; at(4, 8) = -r[T1D_P_P2U] * r[T1D_P_IB];
    at(5, 5) = -r[T1D_P_KI]; at(5, 0) = r[T1D_P_KI] / r[T1D_P_VI];
    at(6, 6) = -r[T1D_P_KI]; at(6, 5) = r[T1D_P_KI];
    double Ah[81], Ph[81], Pk[81], tmp[81];
    const double h = 1.0 / (double)ng;
    for (int k = 0; k < 81; ++k) Ah[k] = A[k] * h;
    mat_expm(9, Ah, Ph);
    std::memcpy(Pk, Ph, sizeof(Pk));
    static const int c6[7] = {4, 0, 1, 2, 3, 7, 8};      // x6 <- x6, x5, x9, x10, x11, u, 1
    static const int c8[7] = {6, 5, 0, 1, 2, 3, 7};      // x8 <- x8, x7, x5, x9, x10, x11, u
    for (int k = 1; k <= ng; ++k) {
        double* o = out + (k - 1) * 14;
        for (int j = 0; j < 7; ++j) { o[j] = Pk[4 * 9 + c6[j]]; o[7 + j] = Pk[6 * 9 + c8[j]]; }
        if (k == ng) break;
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) {
                double v = 0.0;
                for (int q = 0; q < 9; ++q) v += Ph[i * 9 + q] * Pk[q * 9 + j];
                tmp[i * 9 + j] = v;
            }
        std::memcpy(Pk, tmp, sizeof(Pk));
    }
    double* t = out + 14 * ng;                           // tail: Phi(1)
    static const int c5[5] = {0, 1, 2, 3, 7};
    for (int j = 0; j < 5; ++j) { t[j] = Pk[0 * 9 + c5[j]]; t[5 + j] = Pk[1 * 9 + c5[j]]; }
    t[10] = Pk[2 * 9 + 2]; t[11] = Pk[2 * 9 + 7];
    t[12] = Pk[3 * 9 + 2]; t[13] = Pk[3 * 9 + 3]; t[14] = Pk[3 * 9 + 7];
    static const int c7[6] = {5, 0, 1, 2, 3, 7};
    for (int j = 0; j < 6; ++j) t[15 + j] = Pk[5 * 9 + c7[j]];
    for (int part = 0; part < 2; ++part) {               // weights for h, then for the refined step h/2
        const double hh = part ? 0.5 * h : h, z = r[T1D_P_KABS] * hh;
        double I[3];
        for (int k = 0; k < 3; ++k) {
            double term = 1.0, sum = 0.0;                // term = (-z)^j k! / (k + j + 1)!
            for (int q = 1; q <= k + 1; ++q) term /= (double)q;
            for (int q = 1; q <= k; ++q) term *= (double)q;
            for (int j = 0; j < 60; ++j) {
                sum += term;
                term *= -z / (double)(k + j + 2);
                if (std::fabs(term) < 1e-30) break;
            }
            I[k] = sum;
        }
        double* w = out + kPropRows(ng) + 4 * part;
        w[0] = std::exp(-z);
        w[1] = hh * (2.0 * I[2] - 3.0 * I[1] + I[0]);
        w[2] = hh * (-4.0 * I[2] + 4.0 * I[1]);
        w[3] = hh * (2.0 * I[2] - I[1]);
    }
}

extern "C" int t1d_split_tables(const double* patient_row, int n_cols, int n_sub, double* out, int out_len)
{
    if (!patient_row || !out) return fail(T1D_E_INVALID, "t1d_split_tables: NULL argument");
    if (n_cols != T1D_P_NCOLS) return fail(T1D_E_INVALID, "t1d_split_tables: n_cols must be T1D_P_NCOLS (45)");
    if (n_sub < 2 || n_sub > 8 || (n_sub & 1)) return fail(T1D_E_INVALID, "t1d_split_tables: n_sub must be 2, 4, 6 or 8");
    if (out_len < kPropRows(n_sub) + 8) return fail(T1D_E_INVALID, "t1d_split_tables: out_len < 14 n_sub + 29");
    split_tables_row(patient_row, n_sub, out);
    return T1D_OK;
}

// (re)build the device tables of the split integrator for n_sub sub-steps per minute
static int ensure_split(t1d_ctx* c, int ng)
{
    if (c->split_nsub == ng) return T1D_OK;
    T1D_HIP(hipDeviceSynchronize());                     // kernels in flight may still be reading the old tables
    const int rows = kPropRows(ng), npp = c->np_pad;
    std::vector<double> prop((size_t)rows * npp, 0.0), one((size_t)rows + 8);
    for (int j = 0; j < c->np; ++j) {
        split_tables_row(c->ptab.data() + (size_t)j * T1D_P_NCOLS, ng, one.data());
        for (int k = 0; k < rows; ++k) prop[(size_t)k * npp + j] = one[k];
        c->dpar[(size_t)DP_X2E * kMaxPatients + j] = one[rows];
        c->dpar[(size_t)DP_X2WA * kMaxPatients + j] = one[rows + 1];
        c->dpar[(size_t)DP_X2WM * kMaxPatients + j] = one[rows + 2];
        c->dpar[(size_t)DP_X2WB * kMaxPatients + j] = one[rows + 3];
        c->dpar[(size_t)DP_X2E2 * kMaxPatients + j] = one[rows + 4];
        c->dpar[(size_t)DP_X2WA2 * kMaxPatients + j] = one[rows + 5];
        c->dpar[(size_t)DP_X2WM2 * kMaxPatients + j] = one[rows + 6];
        c->dpar[(size_t)DP_X2WB2 * kMaxPatients + j] = one[rows + 7];
    }
    std::vector<float> propf(prop.begin(), prop.end()), dpf(c->dpar.begin(), c->dpar.end());
    (void)hipFree(c->d_prop64); (void)hipFree(c->d_prop32); c->d_prop64 = nullptr; c->d_prop32 = nullptr;
    T1D_HIP(hipMalloc((void**)&c->d_prop64, prop.size() * 8));
    T1D_HIP(hipMalloc((void**)&c->d_prop32, propf.size() * 4));
    T1D_HIP(hipMemcpy(c->d_prop64, prop.data(), prop.size() * 8, hipMemcpyHostToDevice));
    T1D_HIP(hipMemcpy(c->d_prop32, propf.data(), propf.size() * 4, hipMemcpyHostToDevice));
    T1D_HIP(hipMemcpy(c->d_par64, c->dpar.data(), c->dpar.size() * 8, hipMemcpyHostToDevice));
    T1D_HIP(hipMemcpy(c->d_par32, dpf.data(), dpf.size() * 4, hipMemcpyHostToDevice));
    c->split_nsub = ng;
    return T1D_OK;
}

// which integrator a call with n_sub sub-steps uses: the split scheme needs fast math and an even n_sub <= 8
static bool use_split(const t1d_ctx* c, int n_sub)
{
    const bool can = c->math != 0 && n_sub >= 2 && n_sub <= 8 && !(n_sub & 1);
    return can && c->integrator != 0;
}

#if T1D_S1_TRACE
extern "C" int t1d_debug_trace(t1d_ctx* c, long long* out) { return hipMemcpy(out, c->d_trace, 96 * 4 * 64 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
#endif
extern "C" int t1d_abi_version(void) { return T1D_ABI_VERSION; }
extern "C" const char* t1d_last_error(void) { return g_err.c_str(); }

extern "C" int t1d_ctx_create(int hip_device, const double* ptab, int n_patients, int n_cols,
                              const double* sensor_row, const double* pump_row, t1d_ctx** out)
{
    try {
        if (!out) return fail(T1D_E_INVALID, "t1d_ctx_create: out is NULL");
        *out = nullptr;
        if (!ptab || !sensor_row || !pump_row) return fail(T1D_E_INVALID, "t1d_ctx_create: NULL table");
        if (n_cols != T1D_P_NCOLS) return fail(T1D_E_INVALID, "t1d_ctx_create: n_cols must be T1D_P_NCOLS (45)");
        if (n_patients < 1 || n_patients > kMaxPatients)
            return fail(T1D_E_INVALID, "t1d_ctx_create: n_patients must be in [1, 64]");
        const double st = sensor_row[5];
        if (!(st >= 1.0) || st != std::floor(st) || st > 1440.0)
            return fail(T1D_E_INVALID, "t1d_ctx_create: sensor sample_time must be a whole number of minutes >= 1");
        if (st > 150.0) return fail(T1D_E_INVALID, "t1d_ctx_create: sensor sample_time must be <= 150 minutes");
        for (int k = 2; k < 6; k += 3)
            if (!(pump_row[k] > 0.0)) return fail(T1D_E_INVALID, "t1d_ctx_create: pump increments must be > 0");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
            return fail(T1D_E_NODEVICE, "t1d_ctx_create: no HIP device visible");
        if (hip_device < 0 || hip_device >= ndev) return fail(T1D_E_INVALID, "t1d_ctx_create: bad device index");
        T1D_HIP(hipSetDevice(hip_device));
        int n_cu = 0;
        T1D_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, hip_device));

        t1d_ctx* c = new (std::nothrow) t1d_ctx();
        if (!c) return fail(T1D_E_INVALID, "t1d_ctx_create: out of host memory");
        c->device = hip_device; c->n_cu = n_cu > 0 ? n_cu : 256; c->np = n_patients; c->S = (int)std::floor(150.0 / st);   // noise_gen.py:41-42
        std::memcpy(c->sensor, sensor_row, sizeof(c->sensor));
        std::memcpy(c->pump, pump_row, sizeof(c->pump));

        const int np = n_patients;
        std::vector<double> dp((size_t)DP_COUNT * kMaxPatients, 0.0), x0((size_t)13 * np);
        for (int j = 0; j < np; ++j) {
            const double* r = ptab + (size_t)j * n_cols;
            auto set = [&](int idx, double v) { dp[(size_t)idx * kMaxPatients + j] = v; };
            set(DP_KMAX, r[T1D_P_KMAX]); set(DP_KMIN, r[T1D_P_KMIN]); set(DP_KABS, r[T1D_P_KABS]);
            set(DP_HK, (r[T1D_P_KMAX] - r[T1D_P_KMIN]) / 2.0);
            set(DP_B, r[T1D_P_B]); set(DP_D, r[T1D_P_D]);
            set(DP_CAA, 5.0 / 2.0 / (1.0 - r[T1D_P_B])); set(DP_CCC, 5.0 / 2.0 / r[T1D_P_D]);
            set(DP_RATC, r[T1D_P_F] * r[T1D_P_KABS] / r[T1D_P_BW]);
            set(DP_KP1, r[T1D_P_KP1]); set(DP_KP2, r[T1D_P_KP2]); set(DP_KP3, r[T1D_P_KP3]);
            set(DP_FSNC, r[T1D_P_FSNC]); set(DP_KE1, r[T1D_P_KE1]); set(DP_KE2, r[T1D_P_KE2]);
            set(DP_K1, r[T1D_P_K1]); set(DP_K2, r[T1D_P_K2]); set(DP_VM0, r[T1D_P_VM0]);
            set(DP_VMX, r[T1D_P_VMX]); set(DP_KM0, r[T1D_P_KM0]);
            set(DP_M24, r[T1D_P_M2] + r[T1D_P_M4]); set(DP_M1, r[T1D_P_M1]);
            set(DP_KA1, r[T1D_P_KA1]); set(DP_KA2, r[T1D_P_KA2]); set(DP_VI, r[T1D_P_VI]);
            set(DP_P2U, r[T1D_P_P2U]); set(DP_IB, r[T1D_P_IB]); set(DP_KI, r[T1D_P_KI]);
            set(DP_M130, r[T1D_P_M1] + r[T1D_P_M30]); set(DP_M2, r[T1D_P_M2]);
            set(DP_KA1KD, r[T1D_P_KA1] + r[T1D_P_KD]); set(DP_KD, r[T1D_P_KD]); set(DP_KSC, r[T1D_P_KSC]);
            set(DP_INSC, 6000.0 / r[T1D_P_BW]); set(DP_VG, r[T1D_P_VG]); set(DP_IVI, 1.0 / r[T1D_P_VI]); set(DP_IVG, 1.0 / r[T1D_P_VG]);
            set(DP_DK, r[T1D_P_KMAX] - r[T1D_P_KMIN]);
            set(DP_CF, r[T1D_P_F] / r[T1D_P_BW]);
            for (int k = 0; k < 13; ++k) x0[(size_t)k * np + j] = r[T1D_P_X0 + k];
        }
        c->ptab.assign(ptab, ptab + (size_t)np * n_cols);
        c->dpar = dp;
        c->np_pad = (np + 1) & ~1;
        std::vector<float> dpf(dp.begin(), dp.end());
        const std::vector<double> minv = spline_second_derivative_operator();
        std::vector<float> minvf(minv.begin(), minv.end());
        auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
            hipError_t e = hipMalloc(dst, bytes);
            if (e != hipSuccess) return e;
            return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        };
        hipError_t e = hipSuccess;
        if (e == hipSuccess) e = up((void**)&c->d_par64, dp.data(), dp.size() * 8);
        if (e == hipSuccess) e = up((void**)&c->d_par32, dpf.data(), dpf.size() * 4);
        if (e == hipSuccess) e = up((void**)&c->d_x0, x0.data(), x0.size() * 8);
        if (e == hipSuccess) e = up((void**)&c->d_minv64, minv.data(), minv.size() * 8);
        if (e == hipSuccess) e = up((void**)&c->d_minv32, minvf.data(), minvf.size() * 4);
#if T1D_S1_TRACE
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_trace, 96 * 4 * 64 * sizeof(long long));
        if (e == hipSuccess) e = hipMemset(c->d_trace, 0, 96 * 4 * 64 * sizeof(long long));
#endif
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_status, sizeof(int));
        if (e == hipSuccess) e = hipMemset(c->d_status, 0, sizeof(int));
        if (e != hipSuccess) {
            std::string m = std::string("t1d_ctx_create: ") + hipGetErrorString(e);
            t1d_ctx_destroy(c);
            return fail(T1D_E_HIP, m);
        }
        *out = c;
        return T1D_OK;
    } catch (const std::exception& ex) {
        return fail(T1D_E_INVALID, std::string("t1d_ctx_create: ") + ex.what());
    } catch (...) {
        return fail(T1D_E_INVALID, "t1d_ctx_create: unknown exception");
    }
}

extern "C" int t1d_ctx_set_option(t1d_ctx* c, const char* name, int64_t value)
{
    if (!c || !name) return fail(T1D_E_INVALID, "t1d_ctx_set_option: NULL argument");
    if (std::strcmp(name, "math") == 0) {
        if (value != 0 && value != 1) return fail(T1D_E_INVALID, "t1d_ctx_set_option: math must be 0 or 1");
        c->math = (int)value;
        return T1D_OK;
    }
    if (std::strcmp(name, "params_mode") == 0) {
        if (value < -1 || value > 1) return fail(T1D_E_INVALID, "t1d_ctx_set_option: params_mode must be -1, 0 or 1");
        c->params_mode = (int)value;
        return T1D_OK;
    }
    if (std::strcmp(name, "split_refill") == 0) {
        if (value != 0 && value != 1) return fail(T1D_E_INVALID, "t1d_ctx_set_option: split_refill must be 0 or 1");
        c->split_refill = (int)value;
        return T1D_OK;
    }
    if (std::strcmp(name, "pipe_stagger") == 0) {
        if (value < 0 || value > 64) return fail(T1D_E_INVALID, "t1d_ctx_set_option: pipe_stagger out of range");
        c->pipe_stagger = (int)value;
        return T1D_OK;
    }
    if (std::strcmp(name, "pipe_blocks") == 0) {
        if (value < 0 || value > 65535) return fail(T1D_E_INVALID, "t1d_ctx_set_option: pipe_blocks out of range");
        c->pipe_blocks = (int)value;
        return T1D_OK;
    }
    if (std::strcmp(name, "pipeline") == 0) {
        if (value != 0 && value != 1) return fail(T1D_E_INVALID, "t1d_ctx_set_option: pipeline must be 0 or 1");
        c->pipeline = (int)value;
        return T1D_OK;
    }
    if (std::strcmp(name, "adaptive_gut") == 0) {
        if (value != 0 && value != 1) return fail(T1D_E_INVALID, "t1d_ctx_set_option: adaptive_gut must be 0 or 1");
        c->adaptive_gut = (int)value;
        return T1D_OK;
    }
    if (std::strcmp(name, "single_minute_kernel") == 0) {
        if (value != 0 && value != 1) return fail(T1D_E_INVALID, "t1d_ctx_set_option: single_minute_kernel must be 0 or 1");
        c->single_minute_kernel = (int)value;
        return T1D_OK;
    }
    if (std::strcmp(name, "integrator") == 0) {
        if (value < -1 || value > 1) return fail(T1D_E_INVALID, "t1d_ctx_set_option: integrator must be -1 (auto), 0 (rk4) or 1 (split)");
        c->integrator = (int)value;
        return T1D_OK;
    }
    if (std::strcmp(name, "scalar_params") == 0) {
        if (value != 0 && value != 1) return fail(T1D_E_INVALID, "t1d_ctx_set_option: scalar_params must be 0 or 1");
        c->scalar_params = (int)value;
        return T1D_OK;
    }
    return fail(T1D_E_INVALID, std::string("t1d_ctx_set_option: unknown option ") + name);
}

extern "C" int t1d_ctx_destroy(t1d_ctx* c)
{
    if (!c) return T1D_OK;
    (void)hipSetDevice(c->device);
    (void)hipFree(c->d_par64); (void)hipFree(c->d_par32); (void)hipFree(c->d_x0);
    (void)hipFree(c->d_minv64); (void)hipFree(c->d_minv32); (void)hipFree(c->d_status);
    (void)hipFree(c->d_prop64); (void)hipFree(c->d_prop32); (void)hipFree(c->d_trace);
    delete c;
    return T1D_OK;
}

static int check_batch(const char* who, const t1d_ctx* c, const t1d_batch* b, bool need_action)
{
    if (!c) return fail(T1D_E_INVALID, std::string(who) + ": ctx is NULL");
    if (!b) return fail(T1D_E_INVALID, std::string(who) + ": batch is NULL");
    if (b->n < 1 || b->n > (int64_t)1 << 28)
        return fail(T1D_E_INVALID, std::string(who) + ": batch.n out of range");
    if (b->dtype != T1D_F64 && b->dtype != T1D_F32) return fail(T1D_E_INVALID, std::string(who) + ": bad dtype");
    if (!b->x || !b->planned || !b->last_qsto || !b->last_food || !b->t || !b->meta || !b->last_cgm ||
        !b->ar_e || !b->pts || !b->prev_cgm)
        return fail(T1D_E_INVALID, std::string(who) + ": a state pointer is NULL");
    if (!b->cgm || !b->bg || !b->reward || !b->done)
        return fail(T1D_E_INVALID, std::string(who) + ": cgm/bg/reward/done outputs are required");
    if (need_action && !b->basal) return fail(T1D_E_INVALID, std::string(who) + ": basal is NULL");
    if (b->n_meals < 0 || b->n_meals > 65535) return fail(T1D_E_INVALID, std::string(who) + ": n_meals out of range");
    if (b->n_meals > 0 && (!b->meal_time || !b->meal_amt))
        return fail(T1D_E_INVALID, std::string(who) + ": n_meals > 0 but meal table pointer is NULL");
    if (b->n_normals < 0) return fail(T1D_E_INVALID, std::string(who) + ": n_normals < 0");
    if (b->n_normals > 0 && !b->normals) return fail(T1D_E_INVALID, std::string(who) + ": n_normals > 0 but normals is NULL");
    return T1D_OK;
}

template <typename T>
static KArgs<T> make_args(const t1d_ctx* c, const t1d_batch* b, int minutes, int n_sub)
{
    KArgs<T> a;
    a.n = b->n; a.env_offset = b->env_offset; a.seed = b->seed;
    a.x = (T*)b->x; a.planned = (T*)b->planned; a.last_qsto = (T*)b->last_qsto; a.last_food = (T*)b->last_food;
    a.t = b->t; a.meta = b->meta; a.episode = b->episode; a.next_meal = b->next_meal;
    a.last_cgm = (T*)b->last_cgm; a.ar_e = (T*)b->ar_e; a.pts = (T*)b->pts; a.prev_cgm = (T*)b->prev_cgm;
    a.basal = (const T*)b->basal; a.bolus = (const T*)b->bolus; a.cho = (const T*)b->cho;
    a.meal_time = b->meal_time; a.meal_amt = (const T*)b->meal_amt;
    a.normals = b->n_normals > 0 ? (const T*)b->normals : nullptr;
    a.x0_override = (const T*)b->x0_override;
    a.cgm = (T*)b->cgm; a.bg = (T*)b->bg; a.reward = (T*)b->reward; a.done = b->done;
    a.lbgi = (T*)b->lbgi; a.hbgi = (T*)b->hbgi; a.risk = (T*)b->risk; a.meal = (T*)b->meal; a.insulin = (T*)b->insulin;
    a.dpar = sizeof(T) == 8 ? (const T*)c->d_par64 : (const T*)c->d_par32;
    a.x0tab = c->d_x0;
    a.minv = sizeof(T) == 8 ? (const T*)c->d_minv64 : (const T*)c->d_minv32;
    a.status = c->d_status; a.trace = c->d_trace;
    a.sen.pacf = (T)c->sensor[0]; a.sen.gamma = (T)c->sensor[1]; a.sen.lambda = (T)c->sensor[2];
    a.sen.delta = (T)c->sensor[3]; a.sen.xi = (T)c->sensor[4]; a.sen.st = (int)c->sensor[5];
    a.sen.vmin = (T)c->sensor[6]; a.sen.vmax = (T)c->sensor[7];
    a.pump.min_bolus = (T)c->pump[0]; a.pump.max_bolus = (T)c->pump[1]; a.pump.inc_bolus = (T)c->pump[2];
    a.pump.min_basal = (T)c->pump[3]; a.pump.max_basal = (T)c->pump[4]; a.pump.inc_basal = (T)c->pump[5];
    a.np = c->np; a.S = c->S; a.n_meals = b->n_meals; a.n_normals = b->n_normals;
    a.minutes = minutes; a.n_sub = n_sub; a.flags = b->flags; a.stagger = c->pipe_stagger;
    a.prop = sizeof(T) == 8 ? (const T*)c->d_prop64 : (const T*)c->d_prop32;
    a.prop_rows = c->split_nsub ? kPropRows(c->split_nsub) : 0; a.np_pad = c->np_pad;
    return a;
}

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

extern "C" int t1d_reset(t1d_ctx* c, const t1d_batch* b, const uint8_t* mask, int random_init_bg, void* stream)
{
    int rc = check_batch("t1d_reset", c, b, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (b->dtype == T1D_F64)
        hipLaunchKernelGGL(reset_kernel<double>, grid_for(b->n), dim3(kBlock), 0, s, make_args<double>(c, b, 1, 1), mask, random_init_bg);
    else
        hipLaunchKernelGGL(reset_kernel<float>, grid_for(b->n), dim3(kBlock), 0, s, make_args<float>(c, b, 1, 1), mask, random_init_bg);
    T1D_HIP(hipGetLastError());
    return T1D_OK;
}

extern "C" int t1d_step(t1d_ctx* c, const t1d_batch* b, int minutes, int n_sub, void* stream)
{
    int rc = check_batch("t1d_step", c, b, true);
    if (rc) return rc;
    if (minutes < 1 || minutes > 100000) return fail(T1D_E_INVALID, "t1d_step: minutes out of range");
    if (n_sub < 1 || n_sub > 4096) return fail(T1D_E_INVALID, "t1d_step: n_sub out of range");
    hipStream_t s = (hipStream_t)stream;
    // measured at 1 Mi envs, fp64: VGPR parameters + sub-system-wise RK4 cost ~75 us + 78 us/minute, the LDS-
    // parameter kernel ~55 us + 93 us/minute: equal at one minute per launch, VGPR form ahead beyond
    const int pmode = c->params_mode >= 0 ? c->params_mode : 1;
    if (c->integrator == 1 && !use_split(c, n_sub))
        return fail(T1D_E_INVALID, "t1d_step: the split integrator needs math = 1 and n_sub in {2, 4, 6, 8}");
    const bool split = use_split(c, n_sub) && !c->pipeline;
    size_t dyn = 0;
    if (split) {
        rc = ensure_split(c, n_sub);
        if (rc) return rc;
        dyn = (size_t)kPropRows(n_sub) * c->np_pad * (b->dtype == T1D_F64 ? 8 : 4);
        if (dyn > 65536) return fail(T1D_E_INVALID, "t1d_step: split tables exceed 64 KiB of LDS (n_patients x n_sub too large); use integrator 0");
    }
    const int variant = c->math == 0 ? 0 : (split ? (c->adaptive_gut ? ((pmode && b->dtype == T1D_F32) ? 6 : 7) : (pmode ? 4 : 5)) : (((b->flags & T1D_BATCH_WAVE_UNIFORM) && c->scalar_params) ? 2 : (pmode ? 3 : 1)));
    // (the adaptive scheme always takes its parameters from LDS: with them in VGPRs as well it spills)
#define T1D_LAUNCH_STEP(V, TT) hipLaunchKernelGGL((step_kernel<V, TT>), grid_for(b->n), dim3(kBlock), dyn, s, make_args<TT>(c, b, minutes, n_sub))
#define T1D_LAUNCH_PIPE(V, TT) hipLaunchKernelGGL((step_pipe_kernel<V, TT>), pgrid, dim3(kBlock), 0, s, make_args<TT>(c, b, minutes, n_sub))
    const size_t esz = b->dtype == T1D_F64 ? 8 : 4;
    const char* xb = (const char*)b->x;
    const size_t rowb = (size_t)b->n * esz;
    const bool packed = (const char*)b->planned == xb + 13 * rowb && (const char*)b->last_qsto == xb + 14 * rowb &&
                        (const char*)b->last_food == xb + 15 * rowb && (const char*)b->last_cgm == xb + 16 * rowb &&
                        (const char*)b->prev_cgm == xb + 17 * rowb && (const char*)b->pts == xb + 18 * rowb &&
                        b->next_meal && (const char*)b->meta == (const char*)b->t + (size_t)b->n * 4 &&
                        (const char*)b->next_meal == (const char*)b->t + (size_t)b->n * 8 &&
                        (size_t)kPackedRows * rowb < ((size_t)1 << 32);
    if (c->pipeline && (variant >= 1 && variant <= 3) && packed && b->n % kBlock == 0) {
        // persistent grid: as many blocks as stay resident at T1D_WAVES waves per SIMD (4 SIMDs x waves / 4 waves per block)
        const unsigned resident = c->pipe_blocks > 0 ? (unsigned)c->pipe_blocks : (unsigned)c->n_cu * T1D_WAVES;
        const unsigned ntiles = grid_for(b->n).x;
        const dim3 pgrid(ntiles < resident ? ntiles : resident);
        if (b->dtype == T1D_F64) { if (variant == 1) T1D_LAUNCH_PIPE(1, double); else if (variant == 2) T1D_LAUNCH_PIPE(2, double); else T1D_LAUNCH_PIPE(3, double); }
        else { if (variant == 1) T1D_LAUNCH_PIPE(1, float); else if (variant == 2) T1D_LAUNCH_PIPE(2, float); else T1D_LAUNCH_PIPE(3, float); }
        T1D_HIP(hipGetLastError());
        return T1D_OK;
    }
    // At most one CGM sample per launch (minutes <= sample_time): the noise-block refill runs as its own
    // kernel ahead of a step kernel compiled without it, unless the caller vouches that none is due.
    const bool split_refill = variant != 0 && c->split_refill && minutes <= (int)c->sensor[5];
    if (split_refill && !(b->flags & T1D_BATCH_NO_REFILL_DUE)) {
        if (b->dtype == T1D_F64) hipLaunchKernelGGL(refill_kernel<double>, grid_for(b->n), dim3(kBlock), 0, s, make_args<double>(c, b, minutes, n_sub));
        else hipLaunchKernelGGL(refill_kernel<float>, grid_for(b->n), dim3(kBlock), 0, s, make_args<float>(c, b, minutes, n_sub));
    }
    // one simulated minute per launch with the split integrator: the persistent early-store kernel
    if (split_refill && split && minutes == 1 && c->single_minute_kernel && packed && !(b->flags & 0x600)) {
        const int stride = c->np <= 32 ? 32 : 64;
        const size_t esz1 = b->dtype == T1D_F64 ? 8 : 4;
        const size_t dyn1 = (size_t)(DP_COUNT + kPropRows(n_sub)) * stride * esz1;
        if (dyn1 <= 65536) {
            const int nchunks = (int)((b->n + 63) / 64);
            int blocks = c->pipe_blocks > 0 ? c->pipe_blocks : c->n_cu;       // one workgroup of 4 x T1D_S1_WAVES waves per CU
            if (blocks > nchunks) blocks = nchunks;
            const bool reg = pmode != 0 && !c->adaptive_gut;
            const bool extra = b->lbgi || b->hbgi || b->risk || b->meal || b->insulin;
            const bool adapt = c->adaptive_gut != 0;
#define T1D_LAUNCH_S1(R, TT, ST, EX, AD) hipLaunchKernelGGL((step1_kernel<R, TT, ST, EX, AD>), dim3(blocks), dim3(kS1Threads), dyn1, s, make_args<TT>(c, b, minutes, n_sub), nchunks)
#define T1D_S1_BY_EXTRA(R, TT, ST) do { if (adapt) { if (extra) T1D_LAUNCH_S1(R, TT, ST, true, true); else T1D_LAUNCH_S1(R, TT, ST, false, true); } \
                                        else { if (extra) T1D_LAUNCH_S1(R, TT, ST, true, false); else T1D_LAUNCH_S1(R, TT, ST, false, false); } } while (0)
            if (b->dtype == T1D_F64) {
                if (stride == 32) { if (reg) T1D_S1_BY_EXTRA(true, double, 32); else T1D_S1_BY_EXTRA(false, double, 32); }
                else { if (reg) T1D_S1_BY_EXTRA(true, double, 64); else T1D_S1_BY_EXTRA(false, double, 64); }
            } else {
                if (stride == 32) { if (reg) T1D_S1_BY_EXTRA(true, float, 32); else T1D_S1_BY_EXTRA(false, float, 32); }
                else { if (reg) T1D_S1_BY_EXTRA(true, float, 64); else T1D_S1_BY_EXTRA(false, float, 64); }
            }
#undef T1D_S1_BY_EXTRA
#undef T1D_LAUNCH_S1
            T1D_HIP(hipGetLastError());
            return T1D_OK;
        }
    }
#define T1D_LAUNCH_FAST(V, TT) hipLaunchKernelGGL((step_kernel<V, TT, false>), grid_for(b->n), dim3(kBlock), dyn, s, make_args<TT>(c, b, minutes, n_sub))
#define T1D_BY_VARIANT(L, TT) do { switch (variant) { case 0: L(0, TT); break; case 1: L(1, TT); break; case 2: L(2, TT); break; \
                                                      case 3: L(3, TT); break; case 4: L(4, TT); break; case 5: L(5, TT); break; \
                                                      default: L(7, TT); break; } } while (0)
    if (split_refill) {          // variant != 0 here
        if (b->dtype == T1D_F64) { switch (variant) { case 1: T1D_LAUNCH_FAST(1, double); break; case 2: T1D_LAUNCH_FAST(2, double); break; case 3: T1D_LAUNCH_FAST(3, double); break;
                                                      case 4: T1D_LAUNCH_FAST(4, double); break; case 5: T1D_LAUNCH_FAST(5, double); break;
                                                      default: T1D_LAUNCH_FAST(7, double); break; } }
        else { switch (variant) { case 1: T1D_LAUNCH_FAST(1, float); break; case 2: T1D_LAUNCH_FAST(2, float); break; case 3: T1D_LAUNCH_FAST(3, float); break;
                                  case 4: T1D_LAUNCH_FAST(4, float); break; case 5: T1D_LAUNCH_FAST(5, float); break;
                                  case 6: T1D_LAUNCH_FAST(6, float); break; default: T1D_LAUNCH_FAST(7, float); break; } }
    } else if (b->dtype == T1D_F64) {
        T1D_BY_VARIANT(T1D_LAUNCH_STEP, double);
    } else if (variant == 6) {           // fp32 only: with the parameters in VGPRs the adaptive scheme fits there (fp64 spills)
        T1D_LAUNCH_STEP(6, float);
    } else {
        T1D_BY_VARIANT(T1D_LAUNCH_STEP, float);
    }
#undef T1D_LAUNCH_FAST
#undef T1D_BY_VARIANT
#undef T1D_LAUNCH_STEP
#undef T1D_LAUNCH_PIPE
    T1D_HIP(hipGetLastError());
    return T1D_OK;
}

template <typename T>
static PidArgs<T> make_pid(const t1d_pid* p, int n_steps)
{
    PidArgs<T> c;
    c.P = (T)p->P; c.I = (T)p->I; c.D = (T)p->D; c.target = (T)p->target;
    c.integ = (T*)p->integ; c.prev = (T*)p->prev; c.sum_risk = (T*)p->sum_risk;
    c.min_bg = (T*)p->min_bg; c.max_bg = (T*)p->max_bg; c.n_low = p->n_low; c.n_high = p->n_high;
    c.n_steps = n_steps;
    c.kind = 0; c.bb_basal = nullptr; c.bb_cr = nullptr; c.bb_cf = nullptr; c.bb_prev_meal = nullptr;
    c.bg_trace = (T*)p->bg_trace; c.cgm_trace = (T*)p->cgm_trace; c.cho_trace = (T*)p->cho_trace; c.ins_trace = (T*)p->insulin_trace;
    c.trace_row = p->trace_row;
    return c;
}

template <typename T>
static PidArgs<T> make_bb(const t1d_bb* p, int n_steps)
{
    PidArgs<T> c;
    c.P = c.I = c.D = T(0); c.target = (T)p->target;
    c.integ = nullptr; c.prev = nullptr; c.sum_risk = (T*)p->sum_risk;
    c.min_bg = (T*)p->min_bg; c.max_bg = (T*)p->max_bg; c.n_low = p->n_low; c.n_high = p->n_high;
    c.n_steps = n_steps;
    c.kind = 1; c.bb_basal = (const T*)p->basal; c.bb_cr = (const T*)p->cr; c.bb_cf = (const T*)p->cf; c.bb_prev_meal = (T*)p->prev_meal;
    c.bg_trace = (T*)p->bg_trace; c.cgm_trace = (T*)p->cgm_trace; c.cho_trace = (T*)p->cho_trace; c.ins_trace = (T*)p->insulin_trace;
    c.trace_row = p->trace_row;
    return c;
}

template <typename MK64, typename MK32>
static int launch_rollout(const char* who, t1d_ctx* c, const t1d_batch* b, int n_steps, int minutes, int n_sub, void* stream,
                          MK64 mk64, MK32 mk32)
{
    int rc = check_batch(who, c, b, false);
    if (rc) return rc;
    if (b->cho) return fail(T1D_E_INVALID, std::string(who) + ": dense cho is not supported, use the meal table");
    if (n_steps < 1) return fail(T1D_E_INVALID, std::string(who) + ": n_steps < 1");
    if (minutes < 1 || minutes > 100000) return fail(T1D_E_INVALID, std::string(who) + ": minutes out of range");
    if (n_sub < 1 || n_sub > 4096) return fail(T1D_E_INVALID, std::string(who) + ": n_sub out of range");
    hipStream_t s = (hipStream_t)stream;
    const int pmode = c->params_mode >= 0 ? c->params_mode : 1;
    if (c->integrator == 1 && !use_split(c, n_sub))
        return fail(T1D_E_INVALID, std::string(who) + ": the split integrator needs math = 1 and n_sub in {2, 4, 6, 8}");
    const bool split = use_split(c, n_sub);
    size_t dyn = 0;
    if (split) {
        rc = ensure_split(c, n_sub);
        if (rc) return rc;
        dyn = (size_t)kPropRows(n_sub) * c->np_pad * (b->dtype == T1D_F64 ? 8 : 4);
        if (dyn > 65536) return fail(T1D_E_INVALID, std::string(who) + ": split tables exceed 64 KiB of LDS; use integrator 0");
    }
    const int variant = c->math == 0 ? 0 : (split ? (c->adaptive_gut ? 7 : (pmode ? 4 : 5)) : (((b->flags & T1D_BATCH_WAVE_UNIFORM) && c->scalar_params) ? 2 : (pmode ? 3 : 1)));
#define T1D_LAUNCH_ROLL(V, TT, MK) hipLaunchKernelGGL((rollout_pid_kernel<V, TT>), grid_for(b->n), dim3(kBlock), dyn, s, \
                                                      make_args<TT>(c, b, minutes, n_sub), MK())
#define T1D_BY_VARIANT(TT, MK) do { switch (variant) { case 0: T1D_LAUNCH_ROLL(0, TT, MK); break; case 1: T1D_LAUNCH_ROLL(1, TT, MK); break; \
                                                       case 2: T1D_LAUNCH_ROLL(2, TT, MK); break; case 3: T1D_LAUNCH_ROLL(3, TT, MK); break; \
                                                       case 4: T1D_LAUNCH_ROLL(4, TT, MK); break; case 5: T1D_LAUNCH_ROLL(5, TT, MK); break; \
                                                       default: T1D_LAUNCH_ROLL(7, TT, MK); break; } } while (0)
    if (b->dtype == T1D_F64) T1D_BY_VARIANT(double, mk64);
    else T1D_BY_VARIANT(float, mk32);
#undef T1D_BY_VARIANT
#undef T1D_LAUNCH_ROLL
    T1D_HIP(hipGetLastError());
    return T1D_OK;
}

extern "C" int t1d_rollout_pid(t1d_ctx* c, const t1d_batch* b, const t1d_pid* pid, int n_steps, int minutes,
                               int n_sub, void* stream)
{
    if (!pid || !pid->integ || !pid->prev) return fail(T1D_E_INVALID, "t1d_rollout_pid: pid state is NULL");
    return launch_rollout("t1d_rollout_pid", c, b, n_steps, minutes, n_sub, stream,
                          [&] { return make_pid<double>(pid, n_steps); }, [&] { return make_pid<float>(pid, n_steps); });
}

extern "C" int t1d_rollout_bb(t1d_ctx* c, const t1d_batch* b, const t1d_bb* bb, int n_steps, int minutes,
                              int n_sub, void* stream)
{
    if (!bb || !bb->basal || !bb->cr || !bb->cf || !bb->prev_meal)
        return fail(T1D_E_INVALID, "t1d_rollout_bb: basal / cr / cf / prev_meal must be set");
    return launch_rollout("t1d_rollout_bb", c, b, n_steps, minutes, n_sub, stream,
                          [&] { return make_bb<double>(bb, n_steps); }, [&] { return make_bb<float>(bb, n_steps); });
}

extern "C" int t1d_random_meals(int hip_device, uint64_t seed, int64_t env_offset, int64_t n, int dtype, int days,
                                const int32_t* start_minute_of_day, int start_scalar, int32_t* meal_time, void* meal_amt,
                                void* stream)
{
    if (!meal_time || !meal_amt) return fail(T1D_E_INVALID, "t1d_random_meals: output pointer is NULL");
    if (n < 1 || n > (int64_t)1 << 28) return fail(T1D_E_INVALID, "t1d_random_meals: n out of range");
    if (days < 1 || days > 10000) return fail(T1D_E_INVALID, "t1d_random_meals: days out of range");
    if (dtype != T1D_F64 && dtype != T1D_F32) return fail(T1D_E_INVALID, "t1d_random_meals: bad dtype");
    if (!start_minute_of_day && (start_scalar < 0 || start_scalar >= 1440))
        return fail(T1D_E_INVALID, "t1d_random_meals: start minute of day must be in [0, 1440)");
    T1D_HIP(hipSetDevice(hip_device));
    MealSlots ms;
    const double prob[6] = {0.95, 0.3, 0.95, 0.3, 0.95, 0.3};          // scenario_gen.py:38-45
    const double lb[6] = {5, 9, 10, 14, 16, 20}, ub[6] = {9, 10, 14, 16, 20, 23}, mu[6] = {7, 9.5, 12, 15, 18, 21.5};
    const double sd[6] = {60, 30, 60, 30, 60, 30}, amu[6] = {45, 10, 70, 10, 80, 10}, asd[6] = {10, 5, 10, 5, 10, 5};
    for (int k = 0; k < 6; ++k) {
        ms.prob[k] = prob[k]; ms.lb[k] = lb[k] * 60.0; ms.ub[k] = ub[k] * 60.0; ms.mu[k] = mu[k] * 60.0;
        ms.sd[k] = sd[k]; ms.amu[k] = amu[k]; ms.asd[k] = asd[k];
        const double ca = 0.5 * std::erfc(-((ms.lb[k] - ms.mu[k]) / sd[k]) / std::sqrt(2.0));
        const double cb = 0.5 * std::erfc(-((ms.ub[k] - ms.mu[k]) / sd[k]) / std::sqrt(2.0));
        ms.cdf_a[k] = ca; ms.cdf_w[k] = cb - ca;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == T1D_F64)
        hipLaunchKernelGGL(random_meals_kernel<double>, grid_for(n), dim3(kBlock), 0, s, seed, env_offset, n, days,
                           start_minute_of_day, start_scalar, meal_time, (double*)meal_amt, ms);
    else
        hipLaunchKernelGGL(random_meals_kernel<float>, grid_for(n), dim3(kBlock), 0, s, seed, env_offset, n, days,
                           start_minute_of_day, start_scalar, meal_time, (float*)meal_amt, ms);
    T1D_HIP(hipGetLastError());
    return T1D_OK;
}

extern "C" int t1d_outcome_stats(int hip_device, int dtype, int64_t n, int64_t n_rows, const void* bg_trace,
                                 const t1d_outcome* out, void* stream)
{
    if (!bg_trace || !out) return fail(T1D_E_INVALID, "t1d_outcome_stats: NULL argument");
    if (n < 1 || n > (int64_t)1 << 28 || n_rows < 1) return fail(T1D_E_INVALID, "t1d_outcome_stats: n / n_rows out of range");
    if (dtype != T1D_F64 && dtype != T1D_F32) return fail(T1D_E_INVALID, "t1d_outcome_stats: bad dtype");
    if (out->risk_trace && out->chunk < 1) return fail(T1D_E_INVALID, "t1d_outcome_stats: chunk < 1");
    if ((out->pct || out->zone) && !(out->q_lo >= 0.0 && out->q_lo <= 100.0 && out->q_hi >= 0.0 && out->q_hi <= 100.0))
        return fail(T1D_E_INVALID, "t1d_outcome_stats: percentiles must be in [0, 100]");
    T1D_HIP(hipSetDevice(hip_device));
    hipStream_t s = (hipStream_t)stream;
    const int chunk = out->chunk > 0 ? out->chunk : 60;
    if (dtype == T1D_F64)
        hipLaunchKernelGGL(outcome_kernel<double>, grid_for(n), dim3(kBlock), 0, s, n, n_rows, (const double*)bg_trace, out->counts,
                           (double*)out->pct, out->zone, (double*)out->risk_trace, out->q_lo, out->q_hi, chunk);
    else
        hipLaunchKernelGGL(outcome_kernel<float>, grid_for(n), dim3(kBlock), 0, s, n, n_rows, (const float*)bg_trace, out->counts,
                           (float*)out->pct, out->zone, (float*)out->risk_trace, out->q_lo, out->q_hi, chunk);
    T1D_HIP(hipGetLastError());
    return T1D_OK;
}

extern "C" int t1d_philox_normals(t1d_ctx* c, uint64_t seed, int64_t env_offset, int64_t n, uint32_t episode,
                                  int32_t draw0, int32_t n_draws, double* out, void* stream)
{
    if (!c || !out || n < 1 || n_draws < 1 || draw0 < -3) return fail(T1D_E_INVALID, "t1d_philox_normals: bad argument");
    hipLaunchKernelGGL(philox_normals_kernel, grid_for(n), dim3(kBlock), 0, (hipStream_t)stream, seed, env_offset, n,
                       episode, draw0, n_draws, out);
    T1D_HIP(hipGetLastError());
    return T1D_OK;
}

extern "C" int t1d_sync(t1d_ctx* c, void* stream, int32_t* status)
{
    if (!c) return fail(T1D_E_INVALID, "t1d_sync: ctx is NULL");
    T1D_HIP(hipStreamSynchronize((hipStream_t)stream));
    int st = 0;
    T1D_HIP(hipMemcpy(&st, c->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) T1D_HIP(hipMemset(c->d_status, 0, sizeof(int)));
    if (status) *status = st;
    if (st) return fail(T1D_E_STATUS, "t1d_sync: device status bits set: " + std::to_string(st));
    return T1D_OK;
}
