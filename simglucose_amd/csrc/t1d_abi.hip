// t1d_abi.hip -- host side of the C ABI of libt1d_hip.so (gfx950 only; see include/t1d.h): context and tables,
// argument checks, kernel selection and launches.  The kernels are in t1d_kernels.hpp, the per-lane arithmetic in
// t1d_device.hpp.  This is the one translation unit of the library.
#include "../../include/t1d.h"
#include "t1d_kernels.hpp"

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

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
    int n_cu = 256;
    int lds_per_block = 65536;   // hipDeviceAttributeMaxSharedMemoryPerBlock (160 KiB on gfx950)
    std::vector<std::pair<const void*, size_t>> lds_allowed;   // kernels whose dynamic-LDS ceiling has been raised above 64 KiB, and to what
    int s1_blocks = 0;       // > 0: grid of the single-minute kernels (tests exercise many chunks per block)
    int split_refill = 1;    // 1 = noise-block refills run in their own kernel ahead of a refill-free step kernel
    int adaptive_gut = 1;    // 1 (default) = the split integrator picks its step sizes per minute and env; 0 = level 1 everywhere
    int single_minute_kernel = 1;   // 1 = minutes == 1 launches of the split integrator use the persistent early-store kernels
    int integrator = -1;     // 0 = classical RK4 on all 13 states, 1 = split scheme, -1 = split whenever n_sub allows it
    int split_nsub = 0;      // n_sub the split tables on the device were built for (0 = none yet)
    int np_pad = 0;
    double* d_prop64 = nullptr; float* d_prop32 = nullptr;   // [kPropRows(split_nsub)][np_pad]
    long long* d_trace = nullptr;    // T1D_S1_TRACE builds
    int defer_min_chunks = 1;        // adaptive_gut = 1: one-minute launches set lanes of level 2 aside from this many chunks per CU up
    int multi_minute_kernel = 1;     // steps of several minutes (minutes <= sample_time) on the packed layout through the persistent kernel with the state in registers across the minutes: 0 never (generic kernel), 1 = fp64 batches of multi_minute_min_envs envs or more, 2 always
    int multi_minute_min_envs = 262144, multi_minute_min_envs_f32 = 393216;      // measured crossovers: tools/mm_thresholds.py
    int park_cap = 0;                // records for set-aside lanes per workgroup of that kernel (0 = what fits in LDS; tests force the overflow path with a small one)
    int pingpong = 1;                // the persistent one-minute kernels walk a CU's chunks backwards in every other launch (below)
    mutable unsigned launches = 0;   // one-minute launches so far
    int record_group_min = 64;       // the multi-minute kernel: that many waiting records go ahead of a wave's next chunk
    int rollout_launches = 1;        // closed-loop roll-outs as one launch of that kernel per step: 0 never (all steps inside one launch of the generic kernel), 1 from rollout_launches_min_envs envs up, 2 always
    int rollout_launches_min_envs = 524288, rollout_launches_min_envs_f32 = 786432;
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

// One patient row -> kPropRows(n_sub) propagator entries (layout: t1d_device.hpp) followed by the four x2 weights
// E, wa, wm, wb for the gut step of level 1 (h = 1/n_sub) and of level 2 (h/2).  The insulin
// sub-system in the order s = (x5, x9, x10, x11, x6, x7, x8, u, 1) (t1dpatient.py:176-198); weights of
// x2' = -kabs x2 + F (:148) from the moments I_k = int_0^1 exp(-z (1 - s)) s^k ds = sum_j (-z)^j k! / (k + j + 1)!,
// z = kabs h, of the quadratic through F(0), F(h/2), F(h).
static void split_tables_row(const double* r, int n_sub, double* out)
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
    const int nb = 2 * n_sub;                            // blocks: tau = k / nb
    const double hb = 1.0 / (double)nb;
    for (int k = 0; k < 81; ++k) Ah[k] = A[k] * hb;
    mat_expm(9, Ah, Ph);
    std::memcpy(Pk, Ph, sizeof(Pk));
    static const int c6[7] = {4, 0, 1, 2, 3, 7, 8};      // x6 <- x6, x5, x9, x10, x11, u, 1
    static const int c8[7] = {6, 5, 0, 1, 2, 3, 7};      // x8 <- x8, x7, x5, x9, x10, x11, u
    for (int k = 1; k <= nb; ++k) {
        double* o = out + (k - 1) * 14;
        for (int j = 0; j < 7; ++j) { o[j] = Pk[4 * 9 + c6[j]]; o[7 + j] = Pk[6 * 9 + c8[j]]; }
        if (k == nb) break;
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) {
                double v = 0.0;
                for (int q = 0; q < 9; ++q) v += Ph[i * 9 + q] * Pk[q * 9 + j];
                tmp[i * 9 + j] = v;
            }
        std::memcpy(Pk, tmp, sizeof(Pk));
    }
    double* t = out + 14 * nb;                           // tail: Phi(1)
    static const int c5[5] = {0, 1, 2, 3, 7};
    for (int j = 0; j < 5; ++j) { t[j] = Pk[0 * 9 + c5[j]]; t[5 + j] = Pk[1 * 9 + c5[j]]; }
    t[10] = Pk[2 * 9 + 2]; t[11] = Pk[2 * 9 + 7];
    t[12] = Pk[3 * 9 + 2]; t[13] = Pk[3 * 9 + 3]; t[14] = Pk[3 * 9 + 7];
    static const int c7[6] = {5, 0, 1, 2, 3, 7};
    for (int j = 0; j < 6; ++j) t[15 + j] = Pk[5 * 9 + c7[j]];
    const double h1 = 1.0 / (double)n_sub;
    const double hs[2] = {h1, 0.5 * h1};                 // gut step of level 1, 2 (the order of DP_X2E, DP_X2E2)
    for (int part = 0; part < 2; ++part) {
        const double hh = hs[part], z = r[T1D_P_KABS] * hh;
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
        double* w = out + kPropRows(n_sub) + 4 * part;
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
    if (out_len < kPropRows(n_sub) + 8) return fail(T1D_E_INVALID, "t1d_split_tables: out_len < 28 n_sub + 29");
    split_tables_row(patient_row, n_sub, out);
    return T1D_OK;
}

// (re)build the device tables of the split integrator for n_sub sub-steps per minute
static int ensure_split(t1d_ctx* c, int ng)
{
    static_assert(DP_X2WB2 == DP_X2E + 7 && DP_X2E2 == DP_X2E + 4, "x2 weight rows are consecutive");
    if (c->split_nsub == ng) return T1D_OK;
    T1D_HIP(hipDeviceSynchronize());                     // kernels in flight may still be reading the old tables
    const int rows = kPropRows(ng), npp = c->np_pad;
    std::vector<double> prop((size_t)rows * npp, 0.0), one((size_t)rows + 8);
    for (int j = 0; j < c->np; ++j) {
        split_tables_row(c->ptab.data() + (size_t)j * T1D_P_NCOLS, ng, one.data());
        for (int k = 0; k < rows; ++k) prop[(size_t)k * npp + j] = one[k];
        for (int k = 0; k < 8; ++k) c->dpar[(size_t)(DP_X2E + k) * kMaxPatients + j] = one[rows + k];   // DP_X2E .. DP_X2WB2
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
extern "C" int t1d_debug_trace(t1d_ctx* c, long long* out) { return hipMemcpy(out, c->d_trace, 128 * 4 * 64 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
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
        int lds_max = 0;
        if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, hip_device) == hipSuccess && lds_max > 0) c->lds_per_block = lds_max;
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
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_trace, 128 * 4 * 64 * sizeof(long long));
        if (e == hipSuccess) e = hipMemset(c->d_trace, 0, 128 * 4 * 64 * sizeof(long long));
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
    struct Opt { const char* name; int t1d_ctx::*field; int64_t lo, hi; };
    static const Opt opts[] = {
        {"math", &t1d_ctx::math, 0, 1},
        {"split_refill", &t1d_ctx::split_refill, 0, 1},
        {"defer_min_chunks", &t1d_ctx::defer_min_chunks, 0, 65535},
        {"multi_minute_kernel", &t1d_ctx::multi_minute_kernel, 0, 2},
        {"multi_minute_min_envs", &t1d_ctx::multi_minute_min_envs, 0, 1 << 28},
        {"multi_minute_min_envs_f32", &t1d_ctx::multi_minute_min_envs_f32, 0, 1 << 28},
        {"park_cap", &t1d_ctx::park_cap, 0, 65535},
        {"pingpong", &t1d_ctx::pingpong, 0, 1},
        {"record_group_min", &t1d_ctx::record_group_min, 1, 64},
        {"rollout_launches", &t1d_ctx::rollout_launches, 0, 2},
        {"rollout_launches_min_envs", &t1d_ctx::rollout_launches_min_envs, 0, 1 << 28},
        {"rollout_launches_min_envs_f32", &t1d_ctx::rollout_launches_min_envs_f32, 0, 1 << 28},
        {"s1_blocks", &t1d_ctx::s1_blocks, 0, 65535},
        {"adaptive_gut", &t1d_ctx::adaptive_gut, 0, 3},
        {"single_minute_kernel", &t1d_ctx::single_minute_kernel, 0, 1},
        {"integrator", &t1d_ctx::integrator, -1, 1},
    };
    for (const Opt& o : opts)
        if (std::strcmp(name, o.name) == 0) {
            if (value < o.lo || value > o.hi)
                return fail(T1D_E_INVALID, std::string("t1d_ctx_set_option: ") + name + " must be in [" + std::to_string(o.lo) + ", " + std::to_string(o.hi) + "]");
            c->*(o.field) = (int)value;
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
        !b->ar_e || !b->pts || !b->prev_risk)
        return fail(T1D_E_INVALID, std::string(who) + ": a state pointer is NULL");
    if (!b->cgm || !b->bg || !b->reward || !b->done)
        return fail(T1D_E_INVALID, std::string(who) + ": cgm/bg/reward/done outputs are required");
    if (need_action && !b->basal) return fail(T1D_E_INVALID, std::string(who) + ": basal is NULL");
    if (b->n_meals < 0 || b->n_meals > 65535) return fail(T1D_E_INVALID, std::string(who) + ": n_meals out of range");
    if (b->n_meals > 0 && (!b->meal_time || !b->meal_amt))
        return fail(T1D_E_INVALID, std::string(who) + ": n_meals > 0 but meal table pointer is NULL");
    if (b->n_normals < 0) return fail(T1D_E_INVALID, std::string(who) + ": n_normals < 0");
    if (b->n_normals > 0 && !b->normals) return fail(T1D_E_INVALID, std::string(who) + ": n_normals > 0 but normals is NULL");
    const int known = T1D_BATCH_NO_PUMP | T1D_BATCH_NO_REFILL_DUE | (T1D_AB_FLAGS ? 0x1f00 : 0);
    if (b->flags & ~known) return fail(T1D_E_INVALID, std::string(who) + ": unknown bit in batch.flags");
    // a ctx is bound to one device: every entry point that takes one launches (and allocates) there
    if (hipSetDevice(c->device) != hipSuccess) return fail(T1D_E_HIP, std::string(who) + ": hipSetDevice failed");
    return T1D_OK;
}

template <typename T>
static KArgs<T> make_args(const t1d_ctx* c, const t1d_batch* b, int minutes, int n_sub)
{
    KArgs<T> a;
    a.n = b->n; a.env_offset = b->env_offset; a.seed = b->seed;
    a.x = (T*)b->x; a.planned = (T*)b->planned; a.last_qsto = (T*)b->last_qsto; a.last_food = (T*)b->last_food;
    a.t = b->t; a.meta = b->meta; a.episode = b->episode; a.next_meal = b->next_meal;
    a.last_cgm = (T*)b->last_cgm; a.ar_e = (T*)b->ar_e; a.pts = (T*)b->pts; a.prev_risk = (T*)b->prev_risk; a.dbar = (T*)b->dbar;
    a.basal = (const T*)b->basal; a.bolus = (const T*)b->bolus; a.cho = (const T*)b->cho;
    a.meal_time = b->meal_time; a.meal_amt = (const T*)b->meal_amt;
    a.normals = b->n_normals > 0 ? (const T*)b->normals : nullptr;
    a.x0_override = (const T*)b->x0_override;
    a.cgm = (T*)b->cgm; a.bg = (T*)b->bg; a.reward = (T*)b->reward; a.done = b->done;
    a.lbgi = (T*)b->lbgi; a.hbgi = (T*)b->hbgi; a.risk = (T*)b->risk; a.meal = (T*)b->meal; a.insulin = (T*)b->insulin; a.cgm0 = (T*)b->cgm0;
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
    a.minutes = minutes; a.n_sub = n_sub; a.flags = b->flags | (c->pingpong && (c->launches & 1) ? kFlagReverse : 0);
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

// more than 64 KiB of dynamic LDS has to be allowed per kernel (a context belongs to one device)
static hipError_t allow_lds(t1d_ctx* c, const void* fn, size_t bytes)
{
    if (bytes <= 65536) return hipSuccess;
    for (auto& f : c->lds_allowed)
        if (f.first == fn) {
            if (f.second >= bytes) return hipSuccess;
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) f.second = bytes;
            return e;
        }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) c->lds_allowed.push_back({fn, bytes});
    return e;
}

// Which generic kernel variant a call takes (t1d_kernels.hpp): reference arithmetic, fast classical RK4, split at
// level 1, split with per-minute step sizes in place.
static int pick_variant(const t1d_ctx* c, bool split, int dtype)
{
    if (c->math == 0) return 0;
    if (!split) return 3;
    if (!c->adaptive_gut) return 4;
    return dtype == T1D_F32 ? 6 : 7;           // with per-lane step sizes the parameters fit in VGPRs in fp32 only
}

// What the persistent kernels of a call need: the split integrator on the packed layout, the noise refill kept out of
// the step kernel, tables that fit in LDS.
struct PersistPlan {
    bool ok = false;
    int stride = 32, nchunks = 0, blocks = 0, per_block = 0;
    size_t dyn_tables = 0;
};

static bool is_packed(const t1d_batch* b, size_t esz)
{
    const char* xb = (const char*)b->x;
    const size_t rowb = (size_t)b->n * esz;
    return (const char*)b->planned == xb + 13 * rowb && (const char*)b->last_qsto == xb + 14 * rowb &&
           (const char*)b->last_food == xb + 15 * rowb && (const char*)b->last_cgm == xb + 16 * rowb &&
           (const char*)b->prev_risk == xb + 17 * rowb && (const char*)b->pts == xb + 18 * rowb &&
           (const char*)b->dbar == xb + (size_t)kRowDbar * rowb &&
           b->next_meal && (const char*)b->meta == (const char*)b->t + (size_t)b->n * 4 &&
           (const char*)b->next_meal == (const char*)b->t + (size_t)b->n * 8 &&
           (size_t)kPackedRows * rowb < ((size_t)1 << 32);
}

static PersistPlan plan_persistent(const t1d_ctx* c, const t1d_batch* b, int n_sub, bool split, bool split_refill)
{
    PersistPlan p;
    const size_t esz = b->dtype == T1D_F64 ? 8 : 4;
    if (!(split_refill && split && c->single_minute_kernel && is_packed(b, esz)) || (T1D_AB_FLAGS && (b->flags & 0x600))) return p;
    p.stride = c->np <= 32 ? 32 : 64;
    p.dyn_tables = (size_t)(DP_COUNT + kPropRows(n_sub)) * p.stride * esz;
    if (p.dyn_tables + 512 > (size_t)c->lds_per_block) return p;
    p.nchunks = (int)((b->n + 63) / 64);
    p.blocks = c->s1_blocks > 0 ? c->s1_blocks : c->n_cu;           // one workgroup of 4 x T1D_S1_WAVES waves per CU
    if (p.blocks > p.nchunks) p.blocks = p.nchunks;
    p.per_block = (p.nchunks + p.blocks - 1) / p.blocks;
    p.ok = true;
    return p;
}

// LDS of stepn_kernel beside the tables: the redo map (one bit per env of a workgroup's share) and the records -- as many as
// fit, never more than the workgroup's env-minutes.  -> records (a multiple of 64), or -1 where the tables leave no room.
static int stepn_park_cap(const t1d_ctx* c, const PersistPlan& p, size_t esz, int minutes, size_t* dyn)
{
    const size_t rec = (size_t)kSnParkT * esz + (size_t)kSnParkI * sizeof(int);
    const size_t fixed = p.dyn_tables;
    if (p.stride != 32 || fixed + 1024 > (size_t)c->lds_per_block) return -1;
    const size_t map = (size_t)p.per_block * 8 + 8;         // the redo map: one bit per env of the workgroup's share
    if (fixed + map + 1024 > (size_t)c->lds_per_block) return -1;
    long long cap = (long long)(((size_t)c->lds_per_block - 1024 - fixed - map) / rec) / 64 * 64;
    const long long share = ((long long)p.per_block * 64 * minutes + 63) / 64 * 64;
    if (cap > share) cap = share;
    if (c->park_cap > 0 && cap > (c->park_cap + 63) / 64 * 64) cap = (c->park_cap + 63) / 64 * 64;
    *dyn = fixed + (size_t)cap * rec + map;
    return (int)cap;
}

template <typename T>
static PidArgs<T> no_ctrl()
{
    PidArgs<T> c;
    std::memset(&c, 0, sizeof(c));
    return c;
}

// one launch of stepn_kernel: a step of `minutes` minutes, or (CTRL) one closed-loop step
template <typename T, bool EXTRA, bool CTRL>
static int launch_stepn(t1d_ctx* c, const t1d_batch* b, const PersistPlan& p, int cap, size_t dyn, int minutes, int n_sub, const PidArgs<T>& pa, hipStream_t s)
{
    T1D_HIP(allow_lds(c, (const void*)stepn_kernel<T, EXTRA, CTRL>, dyn));
    const int mode = (c->adaptive_gut != 0 ? 1 : 0) | (c->adaptive_gut == 2 ? 2 : 0) | (c->record_group_min << 8);
    ++c->launches;
    hipLaunchKernelGGL((stepn_kernel<T, EXTRA, CTRL>), dim3(p.blocks), dim3(sn_threads<T>()), dyn, s, make_args<T>(c, b, minutes, n_sub), pa,
                       p.nchunks, cap, mode);
    return T1D_OK;
}

template <typename T>
static void launch_refill(const t1d_ctx* c, const t1d_batch* b, int minutes, int n_sub, hipStream_t s)
{
    hipLaunchKernelGGL(refill_kernel<T>, grid_for(b->n), dim3(kBlock), 0, s, make_args<T>(c, b, minutes, n_sub));
}

// the generic kernels keep the propagator table [rows][np_pad] in dynamic LDS (next to a static parameter table in the
// LDS-parameter variants): does it fit?
static bool generic_split_fits(const t1d_ctx* c, int n_sub, size_t esz, size_t* dyn)
{
    *dyn = (size_t)kPropRows(n_sub) * c->np_pad * esz;
    return *dyn + (size_t)DP_COUNT * kMaxPatients * esz + 1024 <= (size_t)c->lds_per_block;
}

extern "C" int t1d_step(t1d_ctx* c, const t1d_batch* b, int minutes, int n_sub, void* stream)
{
    int rc = check_batch("t1d_step", c, b, true);
    if (rc) return rc;
    if (minutes < 1 || minutes > 100000) return fail(T1D_E_INVALID, "t1d_step: minutes out of range");
    if (n_sub < 1 || n_sub > 4096) return fail(T1D_E_INVALID, "t1d_step: n_sub out of range");
    hipStream_t s = (hipStream_t)stream;
    if (c->integrator == 1 && !use_split(c, n_sub))
        return fail(T1D_E_INVALID, "t1d_step: the split integrator needs math = 1 and n_sub in {2, 4, 6, 8}");
    bool split = use_split(c, n_sub);
    const size_t esz = b->dtype == T1D_F64 ? 8 : 4;
    if (split) {
        rc = ensure_split(c, n_sub);
        if (rc) return rc;
    }
    int variant = pick_variant(c, split, b->dtype);
    // At most one CGM sample per launch (minutes <= sample_time): the noise-block refill runs as its own
    // kernel ahead of a step kernel compiled without it, unless the caller vouches that none is due.
    const bool split_refill = variant != 0 && c->split_refill && minutes <= (int)c->sensor[5];
    const PersistPlan p = plan_persistent(c, b, n_sub, split, split_refill);
    size_t dyn_n = 0;
    // (measured, Dexcom steps: fp64 147 against 184 us at 512 Ki envs, 257 against 339 at 1 Mi, 883 against 1254 at 4 Mi, level
    // at 256 Ki, the generic kernel ahead below; fp32 within 6 % of the generic kernel at every size)
    const bool want_n = c->multi_minute_kernel == 2 || (c->multi_minute_kernel == 1 && b->n >= (b->dtype == T1D_F64 ? c->multi_minute_min_envs : c->multi_minute_min_envs_f32));
    const int cap = p.ok && minutes > 1 && want_n ? stepn_park_cap(c, p, esz, minutes, &dyn_n) : -1;
    const bool persistent = p.ok && (minutes == 1 || cap >= 0);
    size_t dyn = 0;
    if (!persistent && split && !generic_split_fits(c, n_sub, esz, &dyn)) {
        // the generic kernel cannot hold this table (many patients x many sub-steps): classical RK4 when the caller left
        // the choice of integrator to the library
        if (c->integrator == 1) return fail(T1D_E_INVALID, "t1d_step: split tables exceed the LDS of a workgroup (n_patients x n_sub too large); use integrator 0 or -1");
        split = false; dyn = 0;
        variant = pick_variant(c, split, b->dtype);
    }
    if (split_refill && !(b->flags & T1D_BATCH_NO_REFILL_DUE)) {
        if (b->dtype == T1D_F64) launch_refill<double>(c, b, minutes, n_sub, s); else launch_refill<float>(c, b, minutes, n_sub, s);
    }
    const bool extra = b->lbgi || b->hbgi || b->risk || b->meal || b->insulin;
    if (persistent && minutes > 1) {
        // a step of several minutes: state in registers across the minutes, lanes of level 2 parked and finished at the end
        if (b->dtype == T1D_F64) rc = extra ? launch_stepn<double, true, false>(c, b, p, cap, dyn_n, minutes, n_sub, no_ctrl<double>(), s)
                                            : launch_stepn<double, false, false>(c, b, p, cap, dyn_n, minutes, n_sub, no_ctrl<double>(), s);
        else rc = extra ? launch_stepn<float, true, false>(c, b, p, cap, dyn_n, minutes, n_sub, no_ctrl<float>(), s)
                        : launch_stepn<float, false, false>(c, b, p, cap, dyn_n, minutes, n_sub, no_ctrl<float>(), s);
        if (rc) return rc;
        T1D_HIP(hipGetLastError());
        return T1D_OK;
    }
    if (persistent) {
        // one simulated minute per launch with the split integrator: the persistent early-store kernels
        const int stride = p.stride, nchunks = p.nchunks, blocks = p.blocks, per_block = p.per_block;
        const size_t dyn1 = p.dyn_tables;
        const bool tiered = c->adaptive_gut != 0;
        // per-minute step sizes: lanes of level 2 set aside and integrated together at the end of the launch
        // (step1d_kernel) where the list of the CU's envs fits next to the tables; adaptive_gut = 2 asks for the
        // in-place form, 3 for the set-aside form at any batch size
        const size_t dyn1d = dyn1 + (size_t)kS1DPark * (18 * esz + 3 * sizeof(int)) + (size_t)per_block * 64 * sizeof(uint16_t);
        const bool defer = tiered && (c->adaptive_gut == 3 || (c->adaptive_gut == 1 && per_block >= c->defer_min_chunks)) &&
                           stride == 32 && per_block * 64 <= 65536 && dyn1d + 512 <= (size_t)c->lds_per_block;
#define T1D_LAUNCH_S1(TT, ST, EX, TI) do { T1D_HIP(allow_lds(c, (const void*)step1_kernel<TT, ST, EX, TI>, dyn1)); \
        hipLaunchKernelGGL((step1_kernel<TT, ST, EX, TI>), dim3(blocks), dim3(s1_threads<TT>()), dyn1, s, make_args<TT>(c, b, minutes, n_sub), nchunks); } while (0)
#define T1D_LAUNCH_S1D(TT, EX) do { T1D_HIP(allow_lds(c, (const void*)step1d_kernel<TT, EX>, dyn1d)); \
        hipLaunchKernelGGL((step1d_kernel<TT, EX>), dim3(blocks), dim3(s1d_threads<TT>()), dyn1d, s, make_args<TT>(c, b, minutes, n_sub), nchunks); } while (0)
#define T1D_S1_BY(TT, ST) do { if (tiered) { if (extra) T1D_LAUNCH_S1(TT, ST, true, true); else T1D_LAUNCH_S1(TT, ST, false, true); } \
                               else { if (extra) T1D_LAUNCH_S1(TT, ST, true, false); else T1D_LAUNCH_S1(TT, ST, false, false); } } while (0)
#define T1D_S1D_BY(TT) do { if (extra) T1D_LAUNCH_S1D(TT, true); else T1D_LAUNCH_S1D(TT, false); } while (0)
        ++c->launches;
        if (defer) {
            if (b->dtype == T1D_F64) T1D_S1D_BY(double); else T1D_S1D_BY(float);
        } else if (b->dtype == T1D_F64) {
            if (stride == 32) T1D_S1_BY(double, 32); else T1D_S1_BY(double, 64);
        } else {
            if (stride == 32) T1D_S1_BY(float, 32); else T1D_S1_BY(float, 64);
        }
#undef T1D_S1D_BY
#undef T1D_S1_BY
#undef T1D_LAUNCH_S1D
#undef T1D_LAUNCH_S1
        T1D_HIP(hipGetLastError());
        return T1D_OK;
    }
#define T1D_LAUNCH_STEP(V, TT, RF) do { T1D_HIP(allow_lds(c, (const void*)step_kernel<V, TT, RF>, dyn)); \
        hipLaunchKernelGGL((step_kernel<V, TT, RF>), grid_for(b->n), dim3(kBlock), dyn, s, make_args<TT>(c, b, minutes, n_sub)); } while (0)
#define T1D_BY_VARIANT(TT, RF, V67) do { switch (variant) { case 0: T1D_LAUNCH_STEP(0, TT, RF); break; case 3: T1D_LAUNCH_STEP(3, TT, RF); break; \
                                                            case 4: T1D_LAUNCH_STEP(4, TT, RF); break; default: T1D_LAUNCH_STEP(V67, TT, RF); break; } } while (0)
    if (split_refill) {          // the refill ran ahead (or none is due): the kernel compiled without it
        if (b->dtype == T1D_F64) T1D_BY_VARIANT(double, false, 7); else T1D_BY_VARIANT(float, false, 6);
    } else {
        if (b->dtype == T1D_F64) T1D_BY_VARIANT(double, true, 7); else T1D_BY_VARIANT(float, true, 6);
    }
#undef T1D_BY_VARIANT
#undef T1D_LAUNCH_STEP
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
    if (c->integrator == 1 && !use_split(c, n_sub))
        return fail(T1D_E_INVALID, std::string(who) + ": the split integrator needs math = 1 and n_sub in {2, 4, 6, 8}");
    bool split = use_split(c, n_sub);
    const size_t esz = b->dtype == T1D_F64 ? 8 : 4;
    if (split) {
        rc = ensure_split(c, n_sub);
        if (rc) return rc;
    }
    int variant = pick_variant(c, split, b->dtype);
    // Large batches: one launch of the persistent multi-minute kernel per step, the controller in its prologue -- the
    // step-size rule's lanes of level 2 are set aside, where the all-steps-in-one-launch kernel runs each wave at the level
    // of its most refined lane.  The noise-block refill goes ahead of every step as in t1d_step (the clocks are the envs').
    const bool per_step = c->rollout_launches == 2 || (c->rollout_launches == 1 && b->n >= (b->dtype == T1D_F64 ? c->rollout_launches_min_envs : c->rollout_launches_min_envs_f32));
    const bool split_refill = variant != 0 && c->split_refill && minutes <= (int)c->sensor[5];
    if (per_step && c->multi_minute_kernel) {       // (a step of one minute too: the kernel takes any minutes >= 1)
        const PersistPlan p = plan_persistent(c, b, n_sub, split, split_refill);
        size_t dyn_n = 0;
        const int cap = p.ok ? stepn_park_cap(c, p, esz, minutes, &dyn_n) : -1;
        if (cap >= 0) {
            for (int k = 0; k < n_steps; ++k) {
                if (b->dtype == T1D_F64) {
                    launch_refill<double>(c, b, minutes, n_sub, s);
                    PidArgs<double> pa = mk64();
                    pa.n_steps = 1; pa.trace_row += k;
                    rc = launch_stepn<double, true, true>(c, b, p, cap, dyn_n, minutes, n_sub, pa, s);
                } else {
                    launch_refill<float>(c, b, minutes, n_sub, s);
                    PidArgs<float> pa = mk32();
                    pa.n_steps = 1; pa.trace_row += k;
                    rc = launch_stepn<float, true, true>(c, b, p, cap, dyn_n, minutes, n_sub, pa, s);
                }
                if (rc) return rc;
            }
            T1D_HIP(hipGetLastError());
            return T1D_OK;
        }
    }
    size_t dyn = 0;
    if (split && !generic_split_fits(c, n_sub, esz, &dyn)) {
        if (c->integrator == 1) return fail(T1D_E_INVALID, std::string(who) + ": split tables exceed the LDS of a workgroup; use integrator 0 or -1");
        split = false; dyn = 0;
        variant = pick_variant(c, split, b->dtype);
    }
#define T1D_LAUNCH_ROLL(V, TT, MK) do { T1D_HIP(allow_lds(c, (const void*)rollout_pid_kernel<V, TT>, dyn)); \
        hipLaunchKernelGGL((rollout_pid_kernel<V, TT>), grid_for(b->n), dim3(kBlock), dyn, s, make_args<TT>(c, b, minutes, n_sub), MK()); } while (0)
#define T1D_BY_VARIANT(TT, MK, V67) do { switch (variant) { case 0: T1D_LAUNCH_ROLL(0, TT, MK); break; case 3: T1D_LAUNCH_ROLL(3, TT, MK); break; \
                                                            case 4: T1D_LAUNCH_ROLL(4, TT, MK); break; default: T1D_LAUNCH_ROLL(V67, TT, MK); break; } } while (0)
    if (b->dtype == T1D_F64) T1D_BY_VARIANT(double, mk64, 7);
    else T1D_BY_VARIANT(float, mk32, 6);
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
    T1D_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(philox_normals_kernel, grid_for(n), dim3(kBlock), 0, (hipStream_t)stream, seed, env_offset, n,
                       episode, draw0, n_draws, out);
    T1D_HIP(hipGetLastError());
    return T1D_OK;
}

extern "C" int t1d_model_rhs(t1d_ctx* c, int dtype, int64_t n, int math, const void* x, const int32_t* pid, const void* cho,
                             const void* insulin, const void* last_qsto, const void* last_food, void* dxdt, void* stream)
{
    if (!c || !x || !pid || !cho || !insulin || !last_qsto || !last_food || !dxdt) return fail(T1D_E_INVALID, "t1d_model_rhs: NULL argument");
    if (n < 1 || n > (int64_t)1 << 28) return fail(T1D_E_INVALID, "t1d_model_rhs: n out of range");
    if (dtype != T1D_F64 && dtype != T1D_F32) return fail(T1D_E_INVALID, "t1d_model_rhs: bad dtype");
    if (math != 0 && math != 1) return fail(T1D_E_INVALID, "t1d_model_rhs: math must be 0 or 1");
    T1D_HIP(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
#define T1D_RHS(M, TT, PAR) hipLaunchKernelGGL((rhs_kernel<M, TT>), grid_for(n), dim3(kBlock), 0, s, n, (const TT*)x, pid, (const TT*)cho, \
                                               (const TT*)insulin, (const TT*)last_qsto, (const TT*)last_food, (TT*)dxdt, (const TT*)PAR, c->np, c->d_status)
    if (dtype == T1D_F64) { if (math) T1D_RHS(1, double, c->d_par64); else T1D_RHS(0, double, c->d_par64); }
    else { if (math) T1D_RHS(1, float, c->d_par32); else T1D_RHS(0, float, c->d_par32); }
#undef T1D_RHS
    T1D_HIP(hipGetLastError());
    return T1D_OK;
}

extern "C" int t1d_sync(t1d_ctx* c, void* stream, int32_t* status)
{
    if (!c) return fail(T1D_E_INVALID, "t1d_sync: ctx is NULL");
    T1D_HIP(hipSetDevice(c->device));
    T1D_HIP(hipStreamSynchronize((hipStream_t)stream));
    int st = 0;
    T1D_HIP(hipMemcpy(&st, c->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) T1D_HIP(hipMemset(c->d_status, 0, sizeof(int)));
    if (status) *status = st;
    if (st) return fail(T1D_E_STATUS, "t1d_sync: device status bits set: " + std::to_string(st));
    return T1D_OK;
}
