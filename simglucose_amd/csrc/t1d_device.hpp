// t1d_device.hpp -- device-side pieces of the fused T1D step for gfx950 (CDNA4, wave64).
//
// One lane = one environment.  The 13-state ODE, the RK4 stages, the pump quantiser, the meal
// bookkeeping, the CGM noise/clamp/hold and the risk/reward epilogue all run in registers; HBM
// sees one coalesced read and one coalesced write of the struct-of-arrays state per launch.
// Per-patient parameters come from a table staged in LDS (param-major, so lanes holding different
// patients hit different banks and lanes holding the same patient broadcast).
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

// ---- derived per-patient constants, one row per patient, stored param-major in LDS ------------
enum DevPar : int {
    DP_KMAX = 0, DP_KMIN, DP_KABS, DP_HK /*(kmax-kmin)/2*/, DP_B, DP_D, DP_CAA /*5/2/(1-b)*/,
    DP_CCC /*5/2/d*/, DP_RATC /*f*kabs/BW*/, DP_KP1, DP_KP2, DP_KP3, DP_FSNC, DP_KE1, DP_KE2,
    DP_K1, DP_K2, DP_VM0, DP_VMX, DP_KM0, DP_M24 /*m2+m4*/, DP_M1, DP_KA1, DP_KA2, DP_VI, DP_P2U,
    DP_IB, DP_KI, DP_M130 /*m1+m30*/, DP_M2, DP_KA1KD /*ka1+kd*/, DP_KD, DP_KSC, DP_INSC /*6000/BW*/,
    DP_VG, DP_COUNT
};
constexpr int kMaxPatients = 64;       // LDS table = DP_COUNT * kMaxPatients * sizeof(T) <= 17.5 KiB
constexpr int kBlock = 256;

template <typename T> struct Pars {
    T kmax, kmin, kabs, hk, b, d, caa, ccc, ratc, kp1, kp2, kp3, fsnc, ke1, ke2, k1, k2, vm0, vmx,
      km0, m24, m1, ka1, ka2, vi, p2u, ib, ki, m130, m2, ka1kd, kd, ksc, insc, vg;
};

template <typename T>
__device__ __forceinline__ Pars<T> load_pars(const T* lds, int np, int pid)
{
    Pars<T> p;
#define T1D_LP(field, idx) p.field = lds[(idx) * np + pid]
    T1D_LP(kmax, DP_KMAX); T1D_LP(kmin, DP_KMIN); T1D_LP(kabs, DP_KABS); T1D_LP(hk, DP_HK);
    T1D_LP(b, DP_B); T1D_LP(d, DP_D); T1D_LP(caa, DP_CAA); T1D_LP(ccc, DP_CCC); T1D_LP(ratc, DP_RATC);
    T1D_LP(kp1, DP_KP1); T1D_LP(kp2, DP_KP2); T1D_LP(kp3, DP_KP3); T1D_LP(fsnc, DP_FSNC);
    T1D_LP(ke1, DP_KE1); T1D_LP(ke2, DP_KE2); T1D_LP(k1, DP_K1); T1D_LP(k2, DP_K2); T1D_LP(vm0, DP_VM0);
    T1D_LP(vmx, DP_VMX); T1D_LP(km0, DP_KM0); T1D_LP(m24, DP_M24); T1D_LP(m1, DP_M1); T1D_LP(ka1, DP_KA1);
    T1D_LP(ka2, DP_KA2); T1D_LP(vi, DP_VI); T1D_LP(p2u, DP_P2U); T1D_LP(ib, DP_IB); T1D_LP(ki, DP_KI);
    T1D_LP(m130, DP_M130); T1D_LP(m2, DP_M2); T1D_LP(ka1kd, DP_KA1KD); T1D_LP(kd, DP_KD);
    T1D_LP(ksc, DP_KSC); T1D_LP(insc, DP_INSC); T1D_LP(vg, DP_VG);
#undef T1D_LP
    return p;
}

// ---- per-minute inputs of the RHS, constant over the RK4 sub-steps (t1dpatient.py:110-111) ----
template <typename T> struct MinuteIn {
    T d_mg;      // eaten CHO, mg/min                      (:121)
    T ins;       // insulin, pmol/kg/min                   (:122)
    T aa, cc;    // tanh slopes                            (:136-137)
    T bD, dD;    // b*Dbar, d*Dbar                         (:139-140)
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

// T1DPatient.model (t1dpatient.py:119-208).  k = dx/dt.
template <typename T>
__device__ __forceinline__ void rhs(const Pars<T>& p, const MinuteIn<T>& u, const T (&x)[13], T (&k)[13])
{
    const T qsto = x[0] + x[1];                                               // :126
    k[0] = -p.kmax * x[0] + u.d_mg;                                            // :133
    const T kg = p.kmin + p.hk * (t_tanh(u.aa * (qsto - u.bD)) - t_tanh(u.cc * (qsto - u.dD)) + T(2)); // :138-140
    const T kgut = u.has_dbar ? kg : p.kmax;                                   // :135,142
    k[1] = p.kmax * x[0] - x[1] * kgut;                                        // :145
    k[2] = kgut * x[1] - p.kabs * x[2];                                        // :148
    const T rat = p.ratc * x[2];                                               // :151
    const T egp = p.kp1 - p.kp2 * x[3] - p.kp3 * x[8];                         // :153
    const T et = (x[3] > p.ke2) ? p.ke1 * (x[3] - p.ke2) : T(0);               // :158-161
    const T d3 = (egp > T(0) ? egp : T(0)) + rat - p.fsnc - et - p.k1 * x[3] + p.k2 * x[4];   // :165
    k[3] = (x[3] >= T(0)) ? d3 : T(0);                                         // :167
    const T vmt = p.vm0 + p.vmx * x[6];                                        // :169
    const T uid = vmt * x[4] / (p.km0 + x[4]);                                 // :171
    const T d4 = -uid + p.k1 * x[3] - p.k2 * x[4];                             // :172
    k[4] = (x[4] >= T(0)) ? d4 : T(0);                                         // :173
    const T d5 = -p.m24 * x[5] + p.m1 * x[9] + p.ka1 * x[10] + p.ka2 * x[11];  // :176
    const T it = x[5] / p.vi;                                                  // :178
    k[5] = (x[5] >= T(0)) ? d5 : T(0);                                         // :179
    k[6] = -p.p2u * x[6] + p.p2u * (it - p.ib);                                // :182
    k[7] = -p.ki * (x[7] - it);                                                // :185
    k[8] = -p.ki * (x[8] - x[7]);                                              // :187
    const T d9 = -p.m130 * x[9] + p.m2 * x[5];                                 // :190
    k[9] = (x[9] >= T(0)) ? d9 : T(0);                                         // :191
    const T d10 = u.ins - p.ka1kd * x[10];                                     // :194
    k[10] = (x[10] >= T(0)) ? d10 : T(0);                                      // :195
    const T d11 = p.kd * x[10] - p.ka2 * x[11];                                // :197
    k[11] = (x[11] >= T(0)) ? d11 : T(0);                                      // :198
    const T d12 = -p.ksc * x[12] + p.ksc * x[3];                               // :201
    k[12] = (x[12] >= T(0)) ? d12 : T(0);                                      // :202
}

// One minute of classical RK4 in n_sub sub-steps; replaces scipy's DOPRI5 (t1dpatient.py:110-113).
template <typename T>
__device__ __forceinline__ void rk4_minute(const Pars<T>& p, const MinuteIn<T>& u, T (&x)[13], int n_sub)
{
    const T h = T(1) / T(n_sub);
    const T hh = T(0.5) * h, h6 = h / T(6);
    T k[13], y[13], acc[13];
    for (int s = 0; s < n_sub; ++s) {
        rhs(p, u, x, k);
#pragma unroll
        for (int i = 0; i < 13; ++i) { acc[i] = k[i]; y[i] = x[i] + hh * k[i]; }
        rhs(p, u, y, k);
#pragma unroll
        for (int i = 0; i < 13; ++i) { acc[i] += T(2) * k[i]; y[i] = x[i] + hh * k[i]; }
        rhs(p, u, y, k);
#pragma unroll
        for (int i = 0; i < 13; ++i) { acc[i] += T(2) * k[i]; y[i] = x[i] + h * k[i]; }
        rhs(p, u, y, k);
#pragma unroll
        for (int i = 0; i < 13; ++i) x[i] = x[i] + h6 * (acc[i] + k[i]);
    }
}

// Meal ingestion bookkeeping of T1DPatient.step (t1dpatient.py:82-107) + _announce_meal (:222-236).
// Returns the MinuteIn for the integrator.
template <typename T>
__device__ __forceinline__ MinuteIn<T> eat_minute(const Pars<T>& p, const T (&x)[13], T meal, T insulin_upm,
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
    u.ins = insulin_upm * p.insc;                           // :122
    const T dbar = last_qsto + last_food * T(1000);         // :130
    u.has_dbar = dbar > T(0);
    const T dsafe = u.has_dbar ? dbar : T(1);
    u.aa = p.caa / dsafe;                                   // :136
    u.cc = p.ccc / dsafe;                                   // :137
    u.bD = p.b * dsafe;
    u.dD = p.d * dsafe;
    return u;
}

// InsulinPump.basal/.bolus (pump.py:23-39); rint = round-half-to-even = np.round.
template <typename T>
__device__ __forceinline__ T pump_quantise(T amount, T inc, T lo, T hi)
{
    T v = amount * T(6000);
    v = t_rint(v / inc) * inc;
    v = v / T(6000);
    v = v < hi ? v : hi;
    v = v > lo ? v : lo;
    return v;
}

// risk_index([bg], 1) (risk.py:5-17); NaN -> 0 and Inf -> max as numpy.nan_to_num does.
template <typename T>
__device__ __forceinline__ void risk_index1(T bg, T& lbgi, T& hbgi, T& ri)
{
    const T f = T(1.509) * (t_pow(t_log(bg), T(1.084)) - T(5.381));
    T l = T(0), h = T(0);
    if (f < T(0)) l = T(10) * f * f;
    if (f > T(0)) h = T(10) * f * f;
    const T big = sizeof(T) == 8 ? T(1.7976931348623157e308) : T(3.4028234663852886e38);
    l = l < big ? l : big;
    h = h < big ? h : big;
    lbgi = l; hbgi = h; ri = l + h;
}

template <typename T> struct SensorC { T pacf, gamma, lambda, delta, xi, vmin, vmax; int st; };
template <typename T> struct PumpC { T min_bolus, max_bolus, inc_bolus, min_basal, max_basal, inc_basal; };

template <typename T>
__device__ __forceinline__ T johnson_su(const SensorC<T>& s, T e)
{
    return s.xi + s.lambda * t_sinh((e - s.gamma) / s.delta);
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
