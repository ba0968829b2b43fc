// fp64 VALU issue-rate probe for gfx950 (tuning aid, not part of the library): which of the instructions in the fp64 minute
// are slower than an FMA?  Candidates: the exp core's range reduction and scaling (v_rndne_f64, v_cvt_i32_f64, v_ldexp_f64)
// against integer forms of the same steps (magic-number rounding: two v_add_f64; scaling: one integer add into the high word),
// v_rcp_f64, fp64 compare + select, v_max_f64.  Each lane runs CH independent chains; one workgroup of 256 * W threads per CU.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_issue tools/ubench/fp64_issue.hip && ./fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CH, int KIND>
__global__ void chains(double* out, int iters, double a, double b)
{
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = a + c + threadIdx.x * 1e-6;
    const double magic = 6755399441055744.0;          // 1.5 * 2^52
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (KIND == 0) x[c] = fma(x[c], b, a);                                      // v_fma_f64
            else if (KIND == 1) x[c] = x[c] * b;                                        // v_mul_f64
            else if (KIND == 2) x[c] = x[c] + b;                                        // v_add_f64
            else if (KIND == 3) x[c] = __builtin_amdgcn_rcp(x[c]) + a;                  // v_rcp_f64 + add
            else if (KIND == 4) x[c] = rint(x[c]) + b;                                  // v_rndne_f64 + add
            else if (KIND == 5) x[c] = (double)(int)x[c] + b;                           // v_cvt_i32_f64 + v_cvt_f64_i32 + add
            else if (KIND == 6) x[c] = ldexp(x[c], (int)u - 8) + b;                     // v_ldexp_f64 + add
            else if (KIND == 7) x[c] = fmax(x[c] * b, a);                               // mul + max
            else if (KIND == 8) x[c] = x[c] > a ? x[c] * b : x[c] + b;                  // cmp + 2 cndmask + mul + add
            else if (KIND == 9) { const double t = x[c] + magic; x[c] = (t - magic) + b; }   // magic-number rint: 2 add (+ add)
            else if (KIND == 10) {                                                      // scaling by integer add into the high word (+ add)
                uint64_t bits = (uint64_t)__double_as_longlong(x[c]);
                bits += (uint64_t)(uint32_t)((int)u - 8) << 52;
                x[c] = __longlong_as_double((long long)bits) + b;
            }
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
    chains<CH, KIND><<<grid, block>>>(d, 16, 1.0, 0.999999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chains<CH, KIND><<<grid, block>>>(d, iters, 1.0, 0.999999);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    double* d; hipMalloc(&d, 8);
    const int iters = 32000;
    const char* names[] = {"fma", "mul", "add", "rcp+add", "rndne+add", "cvt_i32+cvt_f64+add", "ldexp+add", "mul+max", "cmp+2cnd+mul+add", "magic rint (3 add)", "int scale + add"};
    const int ops[] = {1, 1, 1, 2, 2, 3, 2, 2, 5, 3, 2};
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz; cycles per SEQUENCE (the instructions named) per wave per SIMD\n", clk);
#define ROW(K) for (int w = 1; w <= 3; ++w) { \
        float t1 = run<1, K>(w, iters, d), t2 = run<2, K>(w, iters, d), t4 = run<4, K>(w, iters, d); \
        double cyc = (double)clk * 1e3; (void)ops; \
        printf("%-22s waves/SIMD %d:  ILP1 %6.2f  ILP2 %6.2f  ILP4 %6.2f\n", names[K], w, \
               t1 * 1e-3 * cyc / ((double)iters * 1 * w), t2 * 1e-3 * cyc / ((double)iters * 2 * w), \
               t4 * 1e-3 * cyc / ((double)iters * 4 * w)); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10)
    return 0;
}
