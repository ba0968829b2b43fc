// t1d_device.hpp -- device-side pieces of the fused T1D step for gfx950 (CDNA4, wave64).
//
// One lane = one environment.  The 13-state ODE, the RK4 stages, the pump quantiser, the meal
// bookkeeping, the CGM noise/clamp/hold and the risk/reward epilogue all run in registers; HBM
// sees one coalesced read and one coalesced write of the struct-of-arrays state per launch.
//
// Per-patient parameters reach the RHS in one of two ways (template policy):
//   ParsLds / ParsLdsS  the table is staged in LDS (param-major: lanes holding different patients hit
//               different banks, equal patients broadcast) and each evaluation re-reads what it
//               needs, so no parameter occupies a VGPR across the sub-step loops;
//   ParsReg     gathered once per lane into VGPRs (no LDS round trip in the dependent chains).
//
// Reference behaviour restated here (paths relative to the reference checkout):
//   rhs()            simglucose/patient/t1dpatient.py:119-208   T1DPatient.model
//   eat_minute()     simglucose/patient/t1dpatient.py:82-107,222-236
//   pump_quantise()  simglucose/actuator/pump.py:23-39
//   risk_index1()    simglucose/analysis/risk.py:5-17
//   johnson_su()     simglucose/sensor/noise_gen.py:11-12
#pragma once
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <stdint.h>

namespace t1d {

// ---- derived per-patient constants, one row per patient, stored param-major ---------------------
enum DevPar : int {
    DP_KMAX = 0, DP_KMIN, DP_KABS, DP_HK /*(kmax-kmin)/2*/, DP_B, DP_D, DP_CAA /*5/2/(1-b)*/,
    DP_CCC /*5/2/d*/, DP_RATC /*f*kabs/BW*/, DP_KP1, DP_KP2, DP_KP3, DP_FSNC, DP_KE1, DP_KE2,
    DP_K1, DP_K2, DP_VM0, DP_VMX, DP_KM0, DP_M24 /*m2+m4*/, DP_M1, DP_KA1, DP_KA2, DP_VI, DP_P2U,
    DP_IB, DP_KI, DP_M130 /*m1+m30*/, DP_M2, DP_KA1KD /*ka1+kd*/, DP_KD, DP_KSC, DP_INSC /*6000/BW*/,
    DP_VG, DP_IVI /*1/Vi*/, DP_IVG /*1/Vg*/, DP_DK /*kmax-kmin*/,
    // split integrator only; the x2 weights depend on n_sub and are rewritten when it changes: gut step of level 1
    // (h = 1/n_sub) and of level 2 (h/2)
    DP_CF /*f/BW*/, DP_X2E /*exp(-kabs h)*/, DP_X2WA, DP_X2WM, DP_X2WB,
    DP_X2E2, DP_X2WA2, DP_X2WM2, DP_X2WB2, DP_COUNT
};
constexpr int DP_RK4_COUNT = DP_CF;    // rows the classical-RK4 kernels stage
constexpr int kMaxPatients = 64;       // row stride of the table (device and LDS): 47 x 64 x 8 B = 23.5 KiB
constexpr int kBlock = 256;

// parameters re-read from LDS at every use; refresh() makes the base opaque so the compiler
// cannot hoist the reads out of an RHS evaluation and pin them in VGPRs
template <typename T> struct ParsLds {
    static constexpr bool kSplitRk4 = false;   // measured (1 Mi envs, fp64): 93 us/minute unsplit vs 98 split
    const T* base;     // the __shared__ table, [DP_COUNT][kMaxPatients]: one ds_read with an immediate offset per use
    int pid;
    __device__ __forceinline__ T operator()(int idx) const { return base[idx * kMaxPatients + pid]; }
    __device__ __forceinline__ void refresh() { asm volatile("" : "+v"(pid)); }
    __device__ __forceinline__ void pin() {}
    __device__ __forceinline__ void pin_split() {}
};
// the parameters the fast-math RHS reads (everything else is per-minute set-up)
__device__ constexpr int kRhsPars[] = {DP_KMAX, DP_DK, DP_KABS, DP_RATC, DP_KP1, DP_KP2, DP_KP3, DP_FSNC, DP_KE1, DP_KE2,
                                       DP_K1, DP_K2, DP_VM0, DP_VMX, DP_KM0, DP_M24, DP_M1, DP_KA1, DP_KA2, DP_IVI, DP_P2U,
                                       DP_IB, DP_KI, DP_M130, DP_M2, DP_KA1KD, DP_KD, DP_KSC};
// the parameters the split integrator reads inside its loops
// the parameters the split integrator reads inside its loops (the x2 weights by level: kSplitW below)
__device__ constexpr int kSplitPars[] = {DP_KMAX, DP_DK, DP_RATC, DP_KP1, DP_KP2, DP_KP3, DP_FSNC, DP_KE1, DP_KE2, DP_K1, DP_K2,
                                         DP_VM0, DP_VMX, DP_KM0, DP_KSC, DP_CF};
__host__ __device__ constexpr int kSplitW(int level) { return level == 1 ? DP_X2E : DP_X2E2; }
// parameters gathered once per lane from the (L2-resident) table and held in VGPRs for the launch
template <typename T> struct ParsReg {
    static constexpr bool kSplitRk4 = true;    // measured: 75 us/minute split vs 79 unsplit, and far fewer spills around the loop
    T v[DP_COUNT];
    __device__ __forceinline__ T operator()(int idx) const { return v[idx]; }
    __device__ __forceinline__ void refresh() {}
    // make the RHS parameters register-resident HERE (any spill reload happens before this point)
    __device__ __forceinline__ void pin()
    {
#pragma unroll
        for (int k = 0; k < (int)(sizeof(kRhsPars) / sizeof(int)); ++k) asm volatile("" : "+v"(v[kRhsPars[k]]));
    }
    __device__ __forceinline__ void pin_split()
    {
#pragma unroll
        for (int k = 0; k < (int)(sizeof(kSplitPars) / sizeof(int)); ++k) asm volatile("" : "+v"(v[kSplitPars[k]]));
    }
    __device__ __forceinline__ void load(const T* __restrict__ tab, int pid)
    {
#pragma unroll
        for (int k = 0; k < DP_COUNT; ++k) v[k] = tab[k * kMaxPatients + pid];
    }
};
// ---- per-minute inputs of the RHS, constant over the RK4 sub-steps (t1dpatient.py:110-111) ----
template <typename T> struct MinuteIn {
    T d_mg;      // eaten CHO, mg/min                      (:121)
    T ins;       // insulin, pmol/kg/min                   (:122)
    T aa, cc;    // tanh slopes (MATH 0) or twice them (MATH 1)   (:136-137)
    T bD, dD;    // b*Dbar, d*Dbar                         (:139-140)
    T aabD, ccdD;   // aa*bD, cc*dD (MATH 1): the tanh arguments are then one FMA each, aa*qsto - aabD
    bool has_dbar;
};

__device__ __forceinline__ double t_tanh(double v) { return tanh(v); }
__device__ __forceinline__ float t_tanh(float v) { return tanhf(v); }
__device__ __forceinline__ double t_sinh(double v) { return sinh(v); }
__device__ __forceinline__ float t_sinh(float v) { return sinhf(v); }
__device__ __forceinline__ double t_log(double v) { return log(v); }
__device__ __forceinline__ float t_log(float v) { return logf(v); }
__device__ __forceinline__ double t_pow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ float t_pow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double t_rint(double v) { return rint(v); }
__device__ __forceinline__ float t_rint(float v) { return rintf(v); }
__device__ __forceinline__ double t_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float t_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double t_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float t_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double t_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float t_max(float a, float b) { return fmaxf(a, b); }

// ---- fast fp64 math ------------------------------------------------------------------------------
// exp(v) for v <= 350 (large negative arguments underflow to 0 through ldexp): n = rint(v log2 e),
// r = v - n ln2 (two-part), Taylor polynomial of degree DEG on |r| <= 0.347, scale by 2^n.
// DEG 12: truncation 1.7e-16 relative (19 VALU ops; ocml tanh: ~150);  DEG 10: 2.2e-13 (17 ops) -- used for
// the gastric-emptying term, where it moves glucose by < 1e-10 mg/dL.
// A 64-bit literal cannot be an operand of a VOP3 fp64 instruction, so every polynomial coefficient lives in a
// register pair.  LOCAL = true materialises it in scalar registers right where it is used (two s_mov_b32): for
// code that runs once per env-step (risk index, noise) this stops the compiler from hoisting ~20 coefficients out
// of the tile loop into VGPRs that then stay allocated -- or get spilled -- across the integration loops.
template <bool LOCAL>
__device__ __forceinline__ double kc(double c)
{
    if (LOCAL) asm volatile("" : "+s"(c));
    return c;
}

template <int DEG = 12, bool LOCAL = false>
__device__ __forceinline__ double exp_core(double v)
{
    const double n = rint(v * 1.4426950408889634074);
    if (DEG == 8) {
        // one-constant range reduction: |n| (ln2 - fl(ln2)) = |n| 2.3e-17, far below the polynomial's 1.2e-12
        const double r = fma(-n, kc<LOCAL>(6.93147180559945309417e-01), v);
        // degree 8 with the two leading coefficients exactly 1, the rest fitted to the relative error on |r| <= ln2/2
        // (1.2e-12; the degree-10 Taylor form below has 3e-13): what the gastric-emptying term gets, 32 times a minute
        double q = 2.4708212316486418e-05;
        q = fma(q, r, kc<LOCAL>(1.990893403424476e-04));
        q = fma(q, r, kc<LOCAL>(1.3889197248979142e-03));
        q = fma(q, r, kc<LOCAL>(8.333281816073849e-03));
        q = fma(q, r, kc<LOCAL>(4.166666440468533e-02));
        q = fma(q, r, kc<LOCAL>(1.6666666784412462e-01));
        q = fma(q, r, kc<LOCAL>(5.000000000426794e-01));
        q = fma(q, r, 1.0);
        q = fma(q, r, 1.0);
        return ldexp(q, (int)n);
    }
    double r = fma(-n, kc<LOCAL>(6.93147180369123816490e-01), v);
    r = fma(-n, kc<LOCAL>(1.90821492927058770002e-10), r);
    double p;
    if (DEG >= 12) {
        p = 2.08767569878680989792e-09;                // 1/12!
        p = fma(p, r, kc<LOCAL>(2.50521083854417187751e-08));     // 1/11!
        p = fma(p, r, kc<LOCAL>(2.75573192239858906526e-07));     // 1/10!
    } else {
        p = 2.75573192239858906526e-07;                // 1/10!
    }
    p = fma(p, r, kc<LOCAL>(2.75573192239858906526e-06));        // 1/9!
    p = fma(p, r, kc<LOCAL>(2.48015873015873015873e-05));        // 1/8!
    p = fma(p, r, kc<LOCAL>(1.98412698412698412698e-04));        // 1/7!
    p = fma(p, r, kc<LOCAL>(1.38888888888888888889e-03));        // 1/6!
    p = fma(p, r, kc<LOCAL>(8.33333333333333333333e-03));        // 1/5!
    p = fma(p, r, kc<LOCAL>(4.16666666666666666667e-02));        // 1/4!
    p = fma(p, r, kc<LOCAL>(1.66666666666666666667e-01));        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
template <int DEG = 12, bool LOCAL = false>
__device__ __forceinline__ float exp_core(float v) { return __expf(v); }

// a / b for finite, normal b: v_rcp_f64 seed (measured: 4.6e-8 relative), one Newton step (2.2e-15), then
// the quotient with one residual correction (~1 ulp; skips the scale/fixup of the IEEE sequence).
__device__ __forceinline__ double fdiv(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    y = fma(y, fma(-b, y, 1.0), y);
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);
}
// (fp32: v_rcp_f32, 1 ulp, and the product -- __fdividef compiles to the IEEE scale / fmas / fixup sequence, ten instructions)
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
// a / b to ~3e-15 relative: the seed, one Newton step, the product; no residual correction (3 instructions less).
// For the quotients inside the sub-step loops (gastric emptying, insulin-dependent utilisation).
__device__ __forceinline__ double fdiv_loop(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    y = fma(y, fma(-b, y, 1.0), y);
    return a * y;
}
__device__ __forceinline__ float fdiv_loop(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// log(v) for finite v > 0 (fdlibm-style: v = 2^e m, m in [sqrt(1/2), sqrt 2), s = f/(2+f),
// degree-7 even polynomial in s^2); ~35 VALU ops, < 1 ulp.
template <bool LOCAL = false>
__device__ __forceinline__ double log_core(double v)
{
    int e = __builtin_amdgcn_frexp_exp(v);            // v = m 2^e, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(v);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = fdiv(f, 2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, kc<LOCAL>(2.222219843214978396e-01)), kc<LOCAL>(3.999999999940941908e-01));
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, kc<LOCAL>(1.818357216161805012e-01)), kc<LOCAL>(2.857142874366239149e-01)), kc<LOCAL>(6.666666666666735130e-01));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return fma(dk, kc<LOCAL>(6.93147180369123816490e-01), f - (hfsq - fma(s, hfsq + R, dk * kc<LOCAL>(1.90821492927058770002e-10))));
}
template <bool LOCAL = false>
__device__ __forceinline__ float log_core(float v) { return __logf(v); }

// ---- T1DPatient.model (t1dpatient.py:119-208): k = dx/dt ------------------------------------------
// MATH 0: ocml tanh and IEEE divisions, written as the reference writes them.
// MATH 1: tanh(A) - tanh(C) + 2 = 2 + 2 (Ea - Ec) / ((Ea + 1)(Ec + 1)) with Ea = exp(2A), Ec = exp(2C)
//         (one exp each, one shared reciprocal; absolute error <= ~4 ulp of the bracket), hence
//         kgut = kmax + (kmax - kmin) (Ea - Ec) / ((Ea + 1)(Ec + 1)); divisions via fdiv();
//         u.aa / u.cc hold 2 aa / 2 cc (0 when Dbar <= 0: the exps cancel and kgut = kmax, :142).
template <int MATH, typename T, typename P>
__device__ __forceinline__ void rhs(P& p, const MinuteIn<T>& u, const T (&x)[13], T (&k)[13])
{
    p.refresh();
    const T qsto = x[0] + x[1];                                               // :126
    const T kmax = p(DP_KMAX);
    k[0] = -kmax * x[0] + u.d_mg;                                              // :133
    T kgut;
    if (MATH == 0) {
        const T kg = p(DP_KMIN) + p(DP_HK) * (t_tanh(u.aa * (qsto - u.bD)) - t_tanh(u.cc * (qsto - u.dD)) + T(2)); // :138-140
        kgut = u.has_dbar ? kg : kmax;                                         // :135,142
    } else {
        const T hi = sizeof(T) == 8 ? T(350) : T(40), lo = sizeof(T) == 8 ? T(-745) : T(-80);
        const T a2 = t_max(t_min(u.aa * (qsto - u.bD), hi), lo);
        const T c2 = t_max(t_min(u.cc * (qsto - u.dD), hi), lo);
        const T ea = exp_core(a2), ec = exp_core(c2);
        kgut = kmax + p(DP_DK) * fdiv(ea - ec, (ea + T(1)) * (ec + T(1)));
    }
    k[1] = kmax * x[0] - x[1] * kgut;                                          // :145
    k[2] = kgut * x[1] - p(DP_KABS) * x[2];                                    // :148
    const T rat = p(DP_RATC) * x[2];                                           // :151
    const T egp = p(DP_KP1) - p(DP_KP2) * x[3] - p(DP_KP3) * x[8];             // :153
    const T ke2 = p(DP_KE2);
    const T et = MATH == 0 ? ((x[3] > ke2) ? p(DP_KE1) * (x[3] - ke2) : T(0))  // :158-161
                           : p(DP_KE1) * t_max(x[3] - ke2, T(0));
    const T k1x3 = p(DP_K1) * x[3], k2x4 = p(DP_K2) * x[4];
    const T d3 = t_max(egp, T(0)) + rat - p(DP_FSNC) - et - k1x3 + k2x4;       // :165
    k[3] = (x[3] >= T(0)) ? d3 : T(0);                                         // :167
    const T vmt = p(DP_VM0) + p(DP_VMX) * x[6];                                // :169
    const T uid = MATH == 0 ? vmt * x[4] / (p(DP_KM0) + x[4]) : fdiv(vmt * x[4], p(DP_KM0) + x[4]);   // :171
    const T d4 = -uid + k1x3 - k2x4;                                           // :172
    k[4] = (x[4] >= T(0)) ? d4 : T(0);                                         // :173
    const T d5 = -p(DP_M24) * x[5] + p(DP_M1) * x[9] + p(DP_KA1) * x[10] + p(DP_KA2) * x[11];   // :176
    const T it = MATH == 0 ? x[5] / p(DP_VI) : x[5] * p(DP_IVI);               // :178
    k[5] = (x[5] >= T(0)) ? d5 : T(0);                                         // :179
    const T p2u = p(DP_P2U);
    k[6] = -p2u * x[6] + p2u * (it - p(DP_IB));                                // :182
    const T ki = p(DP_KI);
    k[7] = -ki * (x[7] - it);                                                  // :185
    k[8] = -ki * (x[8] - x[7]);                                                // :187
    const T d9 = -p(DP_M130) * x[9] + p(DP_M2) * x[5];                         // :190
    k[9] = (x[9] >= T(0)) ? d9 : T(0);                                         // :191
    const T d10 = u.ins - p(DP_KA1KD) * x[10];                                 // :194
    k[10] = (x[10] >= T(0)) ? d10 : T(0);                                      // :195
    const T d11 = p(DP_KD) * x[10] - p(DP_KA2) * x[11];                        // :197
    k[11] = (x[11] >= T(0)) ? d11 : T(0);                                      // :198
    const T ksc = p(DP_KSC);
    const T d12 = -ksc * x[12] + ksc * x[3];                                   // :201
    k[12] = (x[12] >= T(0)) ? d12 : T(0);                                      // :202
}

// ---- the same RHS in two independent pieces -----------------------------------------------------
// The insulin sub-system (x5..x11) does not depend on the other six states, and those six need only the
// stage values of x6 (insulin action) and x8 (delayed insulin) from it.  Evaluating the classical RK4
// stages sub-system by sub-system is therefore the SAME arithmetic as evaluating the 13-state RHS four
// times -- every k[i] is formed from the same operands -- but only one sub-system's stage vectors are
// alive at a time: 7 (then 6) x {stage input, accumulator, slope} instead of 13 x 3, which is what
// decides the occupancy of this fp64 kernel.
// xi = (x5, x6, x7, x8, x9, x10, x11)
template <typename T, typename P>
__device__ __forceinline__ void rhs_insulin(P& p, T ins, const T (&xi)[7], T (&k)[7])
{
    p.refresh();
    const T x5 = xi[0], x6 = xi[1], x7 = xi[2], x8 = xi[3], x9 = xi[4], x10 = xi[5], x11 = xi[6];
    const T d5 = -p(DP_M24) * x5 + p(DP_M1) * x9 + p(DP_KA1) * x10 + p(DP_KA2) * x11;          // :176
    const T it = x5 * p(DP_IVI);                                                               // :178
    k[0] = (x5 >= T(0)) ? d5 : T(0);                                                           // :179
    const T p2u = p(DP_P2U);
    k[1] = -p2u * x6 + p2u * (it - p(DP_IB));                                                  // :182
    const T ki = p(DP_KI);
    k[2] = -ki * (x7 - it);                                                                    // :185
    k[3] = -ki * (x8 - x7);                                                                    // :187
    const T d9 = -p(DP_M130) * x9 + p(DP_M2) * x5;                                             // :190
    k[4] = (x9 >= T(0)) ? d9 : T(0);                                                           // :191
    const T d10 = ins - p(DP_KA1KD) * x10;                                                     // :194
    k[5] = (x10 >= T(0)) ? d10 : T(0);                                                         // :195
    const T d11 = p(DP_KD) * x10 - p(DP_KA2) * x11;                                            // :197
    k[6] = (x11 >= T(0)) ? d11 : T(0);                                                         // :198
}
// xg = (x0, x1, x2, x3, x4, x12); x6, x8 = the insulin sub-system's values at this stage
template <typename T, typename P>
__device__ __forceinline__ void rhs_glucose(P& p, const MinuteIn<T>& u, const T (&xg)[6], T x6, T x8, T (&k)[6])
{
    p.refresh();
    const T x0 = xg[0], x1 = xg[1], x2 = xg[2], x3 = xg[3], x4 = xg[4], x12 = xg[5];
    const T qsto = x0 + x1;                                                                    // :126
    const T kmax = p(DP_KMAX);
    k[0] = -kmax * x0 + u.d_mg;                                                                // :133
    const T hi = sizeof(T) == 8 ? T(350) : T(40), lo = sizeof(T) == 8 ? T(-745) : T(-80);
    const T a2 = t_max(t_min(u.aa * (qsto - u.bD), hi), lo);
    const T c2 = t_max(t_min(u.cc * (qsto - u.dD), hi), lo);
    const T ea = exp_core(a2), ec = exp_core(c2);
    const T kgut = kmax + p(DP_DK) * fdiv(ea - ec, (ea + T(1)) * (ec + T(1)));                 // :135-142
    k[1] = kmax * x0 - x1 * kgut;                                                              // :145
    k[2] = kgut * x1 - p(DP_KABS) * x2;                                                        // :148
    const T rat = p(DP_RATC) * x2;                                                             // :151
    const T egp = p(DP_KP1) - p(DP_KP2) * x3 - p(DP_KP3) * x8;                                 // :153
    const T et = p(DP_KE1) * t_max(x3 - p(DP_KE2), T(0));                                      // :158-161
    const T k1x3 = p(DP_K1) * x3, k2x4 = p(DP_K2) * x4;
    const T d3 = t_max(egp, T(0)) + rat - p(DP_FSNC) - et - k1x3 + k2x4;                       // :165
    k[3] = (x3 >= T(0)) ? d3 : T(0);                                                           // :167
    const T vmt = p(DP_VM0) + p(DP_VMX) * x6;                                                  // :169
    const T uid = fdiv(vmt * x4, p(DP_KM0) + x4);                                              // :171
    const T d4 = -uid + k1x3 - k2x4;                                                           // :172
    k[4] = (x4 >= T(0)) ? d4 : T(0);                                                           // :173
    const T ksc = p(DP_KSC);
    const T d12 = -ksc * x12 + ksc * x3;                                                       // :201
    k[5] = (x12 >= T(0)) ? d12 : T(0);                                                         // :202
}

// Classical RK4 sub-steps, insulin sub-system first, then the other six states with its stage values.
template <typename T, typename P>
__device__ __forceinline__ void rk4_substeps_split(P& p, const MinuteIn<T>& u, T (&x)[13], int n_sub)
{
    const T h = T(1) / T(n_sub);
    const T hh = T(0.5) * h, h6 = h / T(6);
    for (int s = 0; s < n_sub; ++s) {
        T s6[4], s8[4];                      // x6, x8 as the four stages see them
        {
            T xi[7], y[7], acc[7], k[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) xi[i] = x[5 + i];
            s6[0] = xi[1]; s8[0] = xi[3];
            rhs_insulin(p, u.ins, xi, k);
#pragma unroll
            for (int i = 0; i < 7; ++i) { acc[i] = k[i]; y[i] = xi[i] + hh * k[i]; }
            s6[1] = y[1]; s8[1] = y[3];
            rhs_insulin(p, u.ins, y, k);
#pragma unroll
            for (int i = 0; i < 7; ++i) { acc[i] += T(2) * k[i]; y[i] = xi[i] + hh * k[i]; }
            s6[2] = y[1]; s8[2] = y[3];
            rhs_insulin(p, u.ins, y, k);
#pragma unroll
            for (int i = 0; i < 7; ++i) { acc[i] += T(2) * k[i]; y[i] = xi[i] + h * k[i]; }
            s6[3] = y[1]; s8[3] = y[3];
            rhs_insulin(p, u.ins, y, k);
#pragma unroll
            for (int i = 0; i < 7; ++i) x[5 + i] = xi[i] + h6 * (acc[i] + k[i]);
        }
        // keep the two passes apart: interleaving them for ILP would bring the register pressure back
        __builtin_amdgcn_sched_barrier(0);
        {
            T xg[6], y[6], acc[6], k[6];
            xg[0] = x[0]; xg[1] = x[1]; xg[2] = x[2]; xg[3] = x[3]; xg[4] = x[4]; xg[5] = x[12];
            rhs_glucose(p, u, xg, s6[0], s8[0], k);
#pragma unroll
            for (int i = 0; i < 6; ++i) { acc[i] = k[i]; y[i] = xg[i] + hh * k[i]; }
            rhs_glucose(p, u, y, s6[1], s8[1], k);
#pragma unroll
            for (int i = 0; i < 6; ++i) { acc[i] += T(2) * k[i]; y[i] = xg[i] + hh * k[i]; }
            rhs_glucose(p, u, y, s6[2], s8[2], k);
#pragma unroll
            for (int i = 0; i < 6; ++i) { acc[i] += T(2) * k[i]; y[i] = xg[i] + h * k[i]; }
            rhs_glucose(p, u, y, s6[3], s8[3], k);
#pragma unroll
            for (int i = 0; i < 6; ++i) xg[i] = xg[i] + h6 * (acc[i] + k[i]);
            x[0] = xg[0]; x[1] = xg[1]; x[2] = xg[2]; x[3] = xg[3]; x[4] = xg[4]; x[12] = xg[5];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- split integrator ------------------------------------------------------------------------------
// The same model advanced part by part with what each part needs (oracle: t1d_o_split_minute):
//   insulin  (x5, x9, x10, x11, x6, x7, x8) is linear with the minute's constant infusion (t1dpatient.py:
//            176-198): exact propagator, s(tau) = Phi(tau) [s; u; 1].  The host stores the structurally
//            non-zero entries of Phi(k / (2 n_sub)), k = 1 .. 2 n_sub (PropRows below); the table is staged in LDS.
//   gut      x0, x1 (:133-145): classical RK4, with Q = int kgut x1 by the same quadrature; x2 (:148), whose rate
//            kabs is the fastest of the model, in exponential form x2' = E x2 + wa F1 + wm (F2 + F3)/2 + wb F4;
//            R += (x2 - x2') + dQ = mass absorbed so far.
//   glucose  (x3, x4, x12) (:151-173,201-202): classical RK4 on z3 = x3 - c R with c = f/BW: the rate of
//            appearance enters through x3 = z3 + c R(tau) at the stage times, where X = x6 and XL = x8 come
//            from the propagator.  A step that begins with x3 < 0 holds x3 (every stage of the reference's RHS
//            returns dx3 = 0 there, :167); a step that takes x3 below zero ends at -1e-10, which is where scipy's
//            step-size control leaves it (a fixed step would overshoot by up to ~0.1 mg/kg and keep that).
// Step sizes per minute and env -- the level -- from the state and the rates at the start of the minute (tier_level):
//   1  gut n_sub steps,   glucose n_sub/2     (the fixed-step form of the scheme takes this one in every minute)
//   2  gut 2 n_sub steps, glucose n_sub       an argument of the gastric-emptying tanh pair (:138-140) moves fast through
//      its transition, x3 is about to reach 0 (:167), or insulin action makes the tissue compartment fast (:169-172):
//      ~0.7 % of the env-minutes of RandomScenario days
// Against a tight solve of random env-days: max 9.2e-4 mg/dL (level 1 everywhere: 6.9e-3).
//
// Propagator rows, 2 n_sub blocks of 14 then a tail of 21 (kPropRows(n_sub) = 28 n_sub + 21):
//   block k = 1..2 n_sub at (k-1)*14, tau = k/(2 n_sub):
//                                 x6(tau) <- [x6, x5, x9, x10, x11, u, 1],  x8(tau) <- [x8, x7, x5, x9, x10, x11, u]
//   tail (tau = 1) at 28 n_sub:   x5 <- [x5, x9, x10, x11, u], x9 <- same, x10 <- [x10, u], x11 <- [x10, x11, u],
//                                 x7 <- [x7, x5, x9, x10, x11, u]
__host__ __device__ constexpr int kPropRows(int n_sub) { return 28 * n_sub + 21; }

struct NoProp { static constexpr bool kSplit = false; };
// compact LDS tables with a compile-time row stride (persistent single-minute kernel): every read is one
// ds_read_b64 with an immediate offset
template <typename T, int STRIDE> struct ParsLdsS {
    static constexpr bool kSplitRk4 = false;
    const T* base; int pid;
    __device__ __forceinline__ T operator()(int idx) const { return base[idx * STRIDE + pid]; }
    __device__ __forceinline__ void refresh() { asm volatile("" : "+v"(pid)); }
    __device__ __forceinline__ void pin() {}
    __device__ __forceinline__ void pin_split() {}
};
template <typename T, int STRIDE> struct PropLdsS {
    static constexpr bool kSplit = true;
    const T* base; int pid;
    __device__ __forceinline__ T operator()(int r) const { return base[r * STRIDE + pid]; }
};
// table in LDS, [rows][stride] with the patient index fastest (different patients -> different banks)
template <typename T> struct PropLds {
    static constexpr bool kSplit = true;
    const T* base; int stride; int pid;
    __device__ __forceinline__ T operator()(int r) const { return base[r * stride + pid]; }
};

// kgut(x0 + x1) * x1, the flux out of the stomach's liquid compartment                         (t1dpatient.py:126-145)
template <typename T, typename P>
__device__ __forceinline__ T kgut_flux(P& p, const MinuteIn<T>& u, T q0, T q1)
{
    const T ehi = sizeof(T) == 8 ? T(350) : T(40);     // only overflow needs a guard: exp of a very negative argument is 0
    const T qsto = q0 + q1;
    // fp64: one FMA per argument (the cancellation costs ~3e-14 absolute); fp32 keeps the difference form
    const T a2 = t_min(sizeof(T) == 8 ? (T)fma((double)u.aa, (double)qsto, -(double)u.aabD) : u.aa * (qsto - u.bD), ehi);
    const T c2 = t_min(sizeof(T) == 8 ? (T)fma((double)u.cc, (double)qsto, -(double)u.ccdD) : u.cc * (qsto - u.dD), ehi);
    const T ea = exp_core<8>(a2), ec = exp_core<8>(c2);
    const T kgut = p(DP_KMAX) + p(DP_DK) * fdiv_loop(ea - ec, (ea + T(1)) * (ec + T(1)));
    return kgut * q1;
}

// d(z3, x4, x12)/dt at a stage of the glucose sub-system: y3 = z3 at the stage, cR = c R and cD = c R' there, X = x6,
// XL = x8.  hold: the step began with x3 = x3hold < 0 and keeps it.  q4 (optional) = Vmt / (Km0 + x4).
template <typename T, typename P>
__device__ __forceinline__ void glucose_rhs(P& p, T y3, T y4, T y12, T cR, T cD, T X, T XL, bool hold, T x3hold,
                                            T& d3, T& d4, T& d12, T* q4 = nullptr)
{
    p.refresh();
    const T x3 = hold ? x3hold : y3 + cR;
    const T egp = p(DP_KP1) - p(DP_KP2) * x3 - p(DP_KP3) * XL;                             // :153
    const T et = p(DP_KE1) * t_max(x3 - p(DP_KE2), T(0));                                  // :158-161
    const T k1x3 = p(DP_K1) * x3, k2x4 = p(DP_K2) * y4;
    const T f3 = t_max(egp, T(0)) - p(DP_FSNC) - et - k1x3 + k2x4;                         // :165 without Rat
    d3 = (x3 >= T(0)) ? f3 : -cD;                                                          // :167: dx3 = 0 <=> dz3 = -c R'
    const T vmt = p(DP_VM0) + p(DP_VMX) * X;                                               // :169
    const T q = fdiv_loop(vmt, p(DP_KM0) + y4);
    const T f4 = -q * y4 + k1x3 - k2x4;                                                    // :171-172
    d4 = (y4 >= T(0)) ? f4 : T(0);                                                         // :173
    const T ksc = p(DP_KSC);
    const T f12 = -ksc * y12 + ksc * x3;                                                   // :201
    d12 = (y12 >= T(0)) ? f12 : T(0);                                                      // :202
    if (q4) *q4 = q;
}

// The step-size rule (oracle: o_tier_level, same constants) -> level 2?  F1 = kgut_flux at the start of the minute,
// dx3 = dx3/dt there (rate of appearance included), q4 = Vmt / (Km0 + x4) there.  Per lane, deterministic.
template <typename T, typename P>
__device__ __forceinline__ bool tier_level2(P& p, const MinuteIn<T>& u, const T (&x)[13], T F1, T dx3, T q4)
{
    const T kNear = T(3), kMove = T(4), kKink = T(1), kStiff = T(2);
    // gut: the arguments of the tanh pair and how far they move in this minute by the rate at its start
    const T dq = u.d_mg - F1;                                   // d(qsto)/dt
    const T q0 = x[0] + x[1];
    const T A0 = T(0.5) * u.aa * (q0 - u.bD), dA = T(0.5) * u.aa * dq;      // u.aa, u.cc hold twice the slopes (0 without Dbar)
    const T C0 = T(0.5) * u.cc * (q0 - u.dD), dC = T(0.5) * u.cc * dq;
    const T A1 = A0 + dA, C1 = C0 + dC;
    bool l2 = (fabs(dA) > kMove && (A0 * A1 <= T(0) || t_min(fabs(A0), fabs(A1)) < kNear)) ||
              (fabs(dC) > kMove && (C0 * C1 <= T(0) || t_min(fabs(C0), fabs(C1)) < kNear));
    // x3 about to reach 0 (:167; a held x3 < 0 has nothing ahead)
    const T x3 = x[3], x3e = x3 + dx3;
    l2 = l2 || (x3 >= T(0) && (x3e <= T(0) || t_min(x3, x3e) < kKink * fabs(dx3)));
    // rate of the tissue compartment under the current insulin action (:169-172)
    return l2 || q4 + p(DP_K2) > kStiff;
}

// One minute.  LEVEL 1 / 2: every lane at that level (compile-time step sizes and table rows); LEVEL 0: every lane at its
// own level, `refine` = level 2 -- one loop whose trip count, step sizes, x2 weights and table rows are per-lane values,
// so a wave runs as long as its most refined lane and the others sit out the extra rounds (kernels that keep the state in
// registers across minutes; the single-minute kernel sets level-2 lanes aside instead).
// HAVE_F1: the caller has evaluated kgut_flux at the start of the minute already (for the step-size rule) and hands it
// in.  (Keeping the rule's first glucose stage as well costs three register pairs across the gut steps: spills.)
// HC (levels 1 and 2): the step sizes come from five words the host has formed with the very expressions below (split_step_sizes)
// and the kernel keeps in LDS -- a minute loop otherwise either repeats three IEEE divisions per minute or carries ten
// more registers across everything.
template <typename T> __host__ __device__ inline void split_step_sizes(int nh, T (&c)[5])
{
    const T h = T(1) / T(nh);
    c[0] = h; c[1] = T(0.5) * h; c[2] = h / T(6); c[3] = h + h; c[4] = (h + h) / T(6);
}
template <int LEVEL, typename T, typename P, typename PR, bool HAVE_F1 = false, bool HC = false>
__device__ __forceinline__ void split_level(P& p, const PR& pr, const MinuteIn<T>& u, T (&x)[13], int n_sub, T f1_pre = T(0),
                                            bool refine = false, const T* hc = nullptr)
{
    // (HC at LEVEL 0: hc holds level 1's five words followed by level 2's)
    const bool r2 = LEVEL == 2 || (LEVEL == 0 && refine);        // this lane at level 2
    const int sb = r2 ? 1 : 2;                                    // propagator blocks per glucose half step
    const int nh = r2 ? 2 * n_sub : n_sub;                        // glucose half steps = gut steps in the minute
    const int ns = nh >> 1;
    T h, H, H6, hh, h6;                                           // h: glucose half step = gut step
    if (HC) {
        int z = LEVEL == 0 && r2 ? 5 : 0;
        asm volatile("" : "+v"(z));
        h = hc[z]; hh = hc[z + 1]; h6 = hc[z + 2]; H = hc[z + 3]; H6 = hc[z + 4];
    } else {
        h = T(1) / T(nh);
        H = h + h; H6 = H / T(6); hh = T(0.5) * h; h6 = h / T(6);
    }
    const T wE = r2 ? p(DP_X2E2) : p(DP_X2E), wA = r2 ? p(DP_X2WA2) : p(DP_X2WA);
    const T wM = r2 ? p(DP_X2WM2) : p(DP_X2WM), wB = r2 ? p(DP_X2WB2) : p(DP_X2WB);
    const T s5 = x[5], s6 = x[6], s7 = x[7], s8 = x[8], s9 = x[9], s10 = x[10], s11 = x[11], ui = u.ins;
    T g0 = x[0], g1 = x[1], x2 = x[2], R = T(0);      // R: mass that left x2 through kabs since the minute began
    T x3a = x[3], x4 = x[4], x12 = x[12];             // x3 itself is carried from step to step: a held value stays bit for bit
    T cRa = T(0), cDa = p(DP_RATC) * x2, x6a = s6, x8a = s8;      // c R, c R', X, XL at the start of the glucose step
    auto kgutF = [&](T q0, T q1) -> T { return kgut_flux(p, u, q0, q1); };
    // One RK4 step of (x0, x1) + the exponential update of x2.  `pre(k)` / `post(k)` run before / after stage k's
    // gastric-emptying evaluation: the caller issues the LDS reads of a propagator row in pre() and consumes them
    // in post(), so that their latency hides behind ~50 dependent VALU instructions instead of being waited out.
    auto gut_step = [&](auto&& pre, auto&& post, bool have_f1) {
        p.refresh();
        const T kmax = p(DP_KMAX);
        pre(0);
        T F1;
        if (HAVE_F1 && have_f1) F1 = f1_pre; else F1 = kgutF(g0, g1);
        post(0, F1);
        const T a0 = u.d_mg - kmax * g0, a1 = kmax * g0 - F1;
        T y0 = g0 + hh * a0, y1 = g1 + hh * a1;
        pre(1);
        const T F2 = kgutF(y0, y1);
        post(1, F2);
        const T b0 = u.d_mg - kmax * y0, b1 = kmax * y0 - F2;
        y0 = g0 + hh * b0; y1 = g1 + hh * b1;
        pre(2);
        const T F3 = kgutF(y0, y1);
        post(2, F3);
        const T c0 = u.d_mg - kmax * y0, c1 = kmax * y0 - F3;
        y0 = g0 + h * c0; y1 = g1 + h * c1;
        pre(3);
        const T F4 = kgutF(y0, y1);
        post(3, F4);
        const T e0 = u.d_mg - kmax * y0, e1 = kmax * y0 - F4;
        const T F23 = F2 + F3;
        g0 += h6 * (a0 + T(2) * (b0 + c0) + e0);
        g1 += h6 * (a1 + T(2) * (b1 + c1) + e1);
        const T x2n = wE * x2 + wA * F1 + wM * (T(0.5) * F23) + wB * F4;
        R += (x2 - x2n) + h6 * (F1 + T(2) * F23 + F4);     // d(x2 + R) = kgut x1 dt
        x2 = x2n;
    };
    auto no_pre = [](int) {};
    auto no_post = [](int, T) {};

    for (int s = 0; s < ns; ++s) {
        // the four propagator rows of this glucose step (x6, x8 at its middle and at its end) ride along the four
        // stages of its first gut step: one row at a time (28 table reads in flight at once would cost 56 VGPRs)
        const int rm = ((2 * s + 1) * sb - 1) * 14, re = rm + sb * 14;
        T cf[7], x6m, x8m, x6b, x8b;
        auto pre = [&](int k) {
            const int r = (k < 2 ? rm : re) + 7 * (k & 1);
#pragma unroll
            for (int j = 0; j < 7; ++j) cf[j] = pr(r + j);
            __builtin_amdgcn_sched_barrier(0);           // the reads are issued here, ahead of the stage
        };
        auto post = [&](int k, T F) {
            // pin: the coefficients are waited for, and the dot product formed, only once this stage's F exists
            asm volatile("" : "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]), "+v"(cf[4]), "+v"(cf[5]), "+v"(cf[6]) : "v"(F));
            const T v6 = cf[0] * s6 + cf[1] * s5 + cf[2] * s9 + cf[3] * s10 + cf[4] * s11 + cf[5] * ui + cf[6];
            const T v8 = cf[0] * s8 + cf[1] * s7 + cf[2] * s5 + cf[3] * s9 + cf[4] * s10 + cf[5] * s11 + cf[6] * ui;
            if (k == 0) x6m = v6; else if (k == 1) x8m = v8; else if (k == 2) x6b = v6; else x8b = v8;
        };
        gut_step(pre, post, s == 0);
        const T cRm = p(DP_CF) * R, cDm = p(DP_RATC) * x2;
        gut_step(no_pre, no_post, false);
        const T cRb = p(DP_CF) * R, cDb = p(DP_RATC) * x2;
        const bool hold = x3a < T(0);
        T z3 = x3a - cRa;
        T k3, k4, k12, a3, a4, a12;
        glucose_rhs(p, z3, x4, x12, cRa, cDa, x6a, x8a, hold, x3a, k3, k4, k12);
        a3 = k3; a4 = k4; a12 = k12;
        glucose_rhs(p, z3 + h * k3, x4 + h * k4, x12 + h * k12, cRm, cDm, x6m, x8m, hold, x3a, k3, k4, k12);
        a3 += T(2) * k3; a4 += T(2) * k4; a12 += T(2) * k12;
        glucose_rhs(p, z3 + h * k3, x4 + h * k4, x12 + h * k12, cRm, cDm, x6m, x8m, hold, x3a, k3, k4, k12);
        a3 += T(2) * k3; a4 += T(2) * k4; a12 += T(2) * k12;
        glucose_rhs(p, z3 + H * k3, x4 + H * k4, x12 + H * k12, cRb, cDb, x6b, x8b, hold, x3a, k3, k4, k12);
        z3 += H6 * (a3 + k3); x4 += H6 * (a4 + k4); x12 += H6 * (a12 + k12);
        const T x3b = z3 + cRb;
        x3a = hold ? x3a : (x3b < T(0) ? T(-1e-10) : x3b);
        cRa = cRb; cDa = cDb; x6a = x6b; x8a = x8b;
    }
    const int t0 = 28 * n_sub;
    x[0] = g0; x[1] = g1; x[2] = x2;
    x[3] = x3a; x[4] = x4; x[12] = x12;
    x[6] = x6a; x[8] = x8a;
    x[5] = pr(t0) * s5 + pr(t0 + 1) * s9 + pr(t0 + 2) * s10 + pr(t0 + 3) * s11 + pr(t0 + 4) * ui;
    x[9] = pr(t0 + 5) * s5 + pr(t0 + 6) * s9 + pr(t0 + 7) * s10 + pr(t0 + 8) * s11 + pr(t0 + 9) * ui;
    x[10] = pr(t0 + 10) * s10 + pr(t0 + 11) * ui;
    x[11] = pr(t0 + 12) * s10 + pr(t0 + 13) * s11 + pr(t0 + 14) * ui;
    x[7] = pr(t0 + 15) * s7 + pr(t0 + 16) * s5 + pr(t0 + 17) * s9 + pr(t0 + 18) * s10 + pr(t0 + 19) * s11 + pr(t0 + 20) * ui;
}

// what the step-size rule needs at the start of the minute; F1 is reused by the integration
template <typename T> struct TierPre { T f1; bool level2; };
template <typename T, typename P>
__device__ __forceinline__ TierPre<T> tier_pre(P& p, const MinuteIn<T>& u, const T (&x)[13])
{
    TierPre<T> t;
    t.f1 = kgut_flux(p, u, x[0], x[1]);
    const T cD = p(DP_RATC) * x[2];
    T k3, k4, k12, q4;
    glucose_rhs(p, x[3], x[4], x[12], T(0), cD, x[6], x[8], x[3] < T(0), x[3], k3, k4, k12, &q4);
    t.level2 = tier_level2(p, u, x, t.f1, k3 + cD, q4);           // dx3 = dz3 + c R' (0 while x3 < 0)
    return t;
}

// One minute, step sizes by the rule, every lane taking its own level in place.
template <typename T, typename P, typename PR>
__device__ __forceinline__ void split_minute_tiered(P& p, const PR& pr, const MinuteIn<T>& u, T (&x)[13], int n_sub)
{
    const TierPre<T> t = tier_pre(p, u, x);
    split_level<0, T, P, PR, true>(p, pr, u, x, n_sub, t.f1, t.level2);
}

template <int MATH, typename T, typename P>
__device__ __forceinline__ void rk4_substeps(P& p, const MinuteIn<T>& u, T (&x)[13], int n_sub);

// One minute of classical RK4 in n_sub sub-steps; replaces scipy's DOPRI5 (t1dpatient.py:110-113).
template <int MATH, typename T, typename P>
__device__ __forceinline__ void rk4_minute(P& p, const MinuteIn<T>& u, T (&x)[13], int n_sub)
{
    if (MATH == 1 && P::kSplitRk4) {
        rk4_substeps_split(p, u, x, n_sub);
    } else {
        rk4_substeps<MATH>(p, u, x, n_sub);
    }
}

template <int MATH, typename T, typename P>
__device__ __forceinline__ void rk4_substeps(P& p, const MinuteIn<T>& u, T (&x)[13], int n_sub)
{
    const T h = T(1) / T(n_sub);
    const T hh = T(0.5) * h, h6 = h / T(6);
    T k[13], y[13], acc[13];
    for (int s = 0; s < n_sub; ++s) {
        rhs<MATH>(p, u, x, k);
#pragma unroll
        for (int i = 0; i < 13; ++i) { acc[i] = k[i]; y[i] = x[i] + hh * k[i]; }
        rhs<MATH>(p, u, y, k);
#pragma unroll
        for (int i = 0; i < 13; ++i) { acc[i] += T(2) * k[i]; y[i] = x[i] + hh * k[i]; }
        rhs<MATH>(p, u, y, k);
#pragma unroll
        for (int i = 0; i < 13; ++i) { acc[i] += T(2) * k[i]; y[i] = x[i] + h * k[i]; }
        rhs<MATH>(p, u, y, k);
#pragma unroll
        for (int i = 0; i < 13; ++i) x[i] = x[i] + h6 * (acc[i] + k[i]);
    }
}

// Dbar = last_Qsto + last_foodtaken * 1000 (t1dpatient.py:130), in ONE form everywhere: the state keeps a copy of it (row
// `dbar`) that has to equal what eat_minute forms from the two words, bit for bit
__device__ __forceinline__ double dbar_of(double lq, double lf) { return fma(lf, 1000.0, lq); }
__device__ __forceinline__ float dbar_of(float lq, float lf) { return fmaf(lf, 1000.0f, lq); }

// Meal ingestion bookkeeping of T1DPatient.step (t1dpatient.py:82-107) + _announce_meal (:222-236).
// Returns the MinuteIn for the integrator.
template <int MATH, typename T, typename P>
__device__ __forceinline__ MinuteIn<T> eat_minute(P& p, const T (&x)[13], T meal, T insulin_upm,
                                                  T& planned, T& last_qsto, T& last_food, bool& was_eating)
{
    T to_eat = T(0);
    planned += meal;                                        // :229
    if (planned > T(0)) {                                   // :230-233
        to_eat = planned < T(5) ? planned : T(5);
        planned -= to_eat;
        planned = planned > T(0) ? planned : T(0);
    }
    if (to_eat > T(0) && !was_eating) {                     // :88-92
        last_qsto = x[0] + x[1];
        last_food = T(0);
    }
    last_food += to_eat;                                    // :98-99 (is_eating <=> to_eat > 0 here)
    was_eating = to_eat > T(0);                             // :102-107
    MinuteIn<T> u;
    u.d_mg = to_eat * T(1000);                              // :121
    u.ins = insulin_upm * p(DP_INSC);                       // :122
    const T dbar = dbar_of(last_qsto, last_food);           // :130
    u.has_dbar = dbar > T(0);
    const T dsafe = u.has_dbar ? dbar : T(1);
    if (MATH == 0) {
        u.aa = u.has_dbar ? p(DP_CAA) / dsafe : T(0);       // :136
        u.cc = u.has_dbar ? p(DP_CCC) / dsafe : T(0);       // :137
    } else {
        const T inv2 = u.has_dbar ? fdiv(T(2), dsafe) : T(0);
        u.aa = p(DP_CAA) * inv2;
        u.cc = p(DP_CCC) * inv2;
    }
    u.bD = p(DP_B) * dsafe;
    u.dD = p(DP_D) * dsafe;
    u.aabD = u.aa * u.bD;
    u.ccdD = u.cc * u.dD;
    return u;
}

// InsulinPump.basal/.bolus (pump.py:23-39); rint = round-half-to-even = np.round.  IEEE divisions
// on purpose: the quantiser must land on the same increment as the reference at exact ties.
template <typename T>
__device__ __forceinline__ T pump_quantise(T amount, T inc, T lo, T hi)
{
    T v = amount * T(6000);
    v = t_rint(v / inc) * inc;          // IEEE division: decides which increment an exact tie rounds to
    v = fdiv(v, T(6000));               // <= 1 ulp: scaling back to U/min decides nothing
    v = v < hi ? v : hi;
    v = v > lo ? v : lo;
    return v;
}

// risk_index([bg], 1) (risk.py:5-17); NaN -> 0 and Inf -> max as numpy.nan_to_num does.
// MATH 1: log(bg)**1.084 = exp(1.084 log(log bg)) with the fast log/exp above; arguments outside
// (1, inf) are resolved by case analysis with numpy's semantics.
template <int MATH, typename T>
__device__ __forceinline__ void risk_index1(T bg, T& lbgi, T& hbgi, T& ri)
{
    T f;
    if (MATH == 0) {
        f = T(1.509) * (t_pow(t_log(bg), T(1.084)) - T(5.381));
    } else {
        // straight-line on purpose (selects, no branches): independent evaluations interleave and hide the
        // latency of the log -> log -> exp chain
        const T inf = T(__builtin_huge_val());
        const T nan = T(__builtin_nan(""));
        const bool regular = bg > T(1) && bg < inf;            // log(bg) in (0, inf)
        const T u = log_core<true>(regular ? bg : T(2));
        const T pw = exp_core<12, true>(T(1.084) * log_core<true>(u));
        const T freg = T(1.509) * (pw - T(5.381));
        // numpy: log(1)**p = 0; log(0) = -inf and (-inf)**p = +inf; log(inf)**p = inf; log of a
        // negative or of a value in (0,1) raised to a non-integer power, and NaN, give NaN
        const T fodd = bg == T(1) ? T(1.509) * (T(0) - T(5.381)) : ((bg == T(0) || bg == inf) ? inf : nan);
        f = regular ? freg : fodd;
    }
    T l = T(0), h = T(0);
    const T ff = T(10) * f * f;
    l = f < T(0) ? ff : T(0);
    h = f > T(0) ? ff : T(0);
    const T big = sizeof(T) == 8 ? T(1.7976931348623157e308) : T(3.4028234663852886e38);
    l = l < big ? l : big;
    h = h < big ? h : big;
    lbgi = l; hbgi = h; ri = l + h;
}

template <typename T> struct SensorC { T pacf, gamma, lambda, delta, xi, vmin, vmax; int st; };
template <typename T> struct PumpC { T min_bolus, max_bolus, inc_bolus, min_basal, max_basal, inc_basal; };

// johnson_transform_SU (noise_gen.py:11-12).  FAST: sinh(v) = (E - 1/E)/2 with E = exp_core(v); the
// cancellation near v = 0 costs relative, not absolute, accuracy (|error| <= ~2e-16 lambda mg/dL).
template <bool FAST, typename T>
__device__ __forceinline__ T johnson_su(const SensorC<T>& s, T e)
{
    const T v = (e - s.gamma) / s.delta;
    if (!FAST) return s.xi + s.lambda * t_sinh(v);
    const T lim = sizeof(T) == 8 ? T(350) : T(40);
    const T E = exp_core(t_max(t_min(v, lim), -lim));
    return s.xi + s.lambda * (T(0.5) * (E - fdiv(T(1), E)));
}

// ---- Philox stream layout ---------------------------------------------------------------------
// subsequence = global env id; one "pair" = one Philox4x32-10 block = two Box-Muller normals.
// pair index inside an episode: 0 = AR(1) initial draw (.x), 1-2 = random_init_bg (3 normals),
// 3 + 5*b + j = the ten normals of noise block b.  Episodes are separated by a 2^24-pair stride.
__device__ __forceinline__ double2 philox_pair(uint64_t seed, uint64_t gid, uint32_t episode, uint32_t pair)
{
    rocrand_state_philox4x32_10 st;
    const unsigned long long off = 4ull * (((unsigned long long)episode << 24) + pair);
    rocrand_init(seed, gid, off, &st);
    return rocrand_normal_double2(&st);
}

} // namespace t1d
