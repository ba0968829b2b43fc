// fp64 VALU issue-rate / latency probe for gfx950 (tuning aid, not part of the library).
// Each lane runs CH independent dependent-FMA chains for `iters` steps; blocks of 64*W threads, one per CU slot.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int KIND>
__global__ void chains(double* out, int iters, double a, double b)
{
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = a + c + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (KIND == 0) x[c] = fma(x[c], b, a);              // v_fma_f64
            else if (KIND == 1) x[c] = x[c] * b;                // v_mul_f64
            else if (KIND == 2) x[c] = x[c] + b;                // v_add_f64
            else if (KIND == 3) x[c] = __builtin_amdgcn_rcp(x[c]) + a;   // v_rcp_f64 + add
            else if (KIND == 4) x[c] = rint(x[c] * b);          // mul + rndne
            else if (KIND == 5) x[c] = fmax(x[c] * b, a);       // mul + max
            else if (KIND == 6) x[c] = ldexp(x[c], 1) * b;      // ldexp + mul
            else if (KIND == 7) x[c] = fma(x[c], x[(c + 1) % CH], x[(c + 2) % CH]);   // three VGPR operands
            else if (KIND == 8) x[c] = fma(x[c], x[(c + 1) % CH], a);                 // one SGPR operand
            else if (KIND == 9) x[c] = fma(x[c], x[(c + 1) % CH], 0.5);               // inline constant
        }
      }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    if (s == 12345.678) out[0] = s;
}
template <int CH, int KIND>
static float run(int waves_per_simd, int iters, double* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256), block(256 * waves_per_simd);       // one workgroup per CU: waves_per_simd waves on each SIMD
    chains<CH, KIND><<<grid, block>>>(d, 10, 1.0, 0.999999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chains<CH, KIND><<<grid, block>>>(d, iters, 1.0, 0.999999);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
__global__ void rcp_err(const double* x, double* err, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double b = x[i];
    const double y0 = __builtin_amdgcn_rcp(b);
    const double ex = 1.0 / b;
    double y1 = fma(y0, fma(-b, y0, 1.0), y0);
    err[i] = fabs(y0 - ex) / fabs(ex);
    err[n + i] = fabs(y1 - ex) / fabs(ex);
}
int main()
{
    {
        const int n = 1 << 20;
        double *hx = new double[n], *he = new double[2 * n], *dx, *de;
        unsigned long long st = 88172645463325252ull;
        for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; hx[i] = ldexp(1.0 + (double)(st >> 11) / 9007199254740992.0, (int)(st % 600) - 300); }
        hipMalloc(&dx, n * 8); hipMalloc(&de, 2 * n * 8); hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
        rcp_err<<<n / 256, 256>>>(dx, de, n); hipMemcpy(he, de, 2 * n * 8, hipMemcpyDeviceToHost);
        double m0 = 0, m1 = 0; for (int i = 0; i < n; ++i) { if (he[i] > m0) m0 = he[i]; if (he[n + i] > m1) m1 = he[n + i]; }
        printf("v_rcp_f64 max relative error %.3e ; after one Newton step %.3e\n", m0, m1);
    }
    double* d; hipMalloc(&d, 8);
    const int iters = 32000;
    const char* names[] = {"fma", "mul", "add", "rcp+add", "mul+rndne", "mul+max", "ldexp+mul", "fma3v", "fma1s", "fma_inl"};
    const int ops[] = {1, 1, 1, 2, 2, 2, 2, 1, 1, 1};
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk);
#define ROW(K) for (int w = 1; w <= 4; ++w) { \
        float t1 = run<1, K>(w, iters, d), t2 = run<2, K>(w, iters, d), t4 = run<4, K>(w, iters, d); \
        double cyc = (double)clk * 1e3; \
        printf("%-10s waves/SIMD %d: cycles per wave-instr (per SIMD)  ILP1 %.2f  ILP2 %.2f  ILP4 %.2f\n", names[K], w, \
               t1 * 1e-3 * cyc / ((double)iters * 1 * ops[K] * w), t2 * 1e-3 * cyc / ((double)iters * 2 * ops[K] * w), \
               t4 * 1e-3 * cyc / ((double)iters * 4 * ops[K] * w)); }
    ROW(0) ROW(7) ROW(8) ROW(9)
    return 0;
}
