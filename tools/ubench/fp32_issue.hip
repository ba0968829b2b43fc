// fp32 VALU issue-rate probe for gfx950 (tuning aid, not part of the library): does two-envs-per-lane packed math
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) buy anything over one env per lane on a SIMD-32 machine?
// Each lane runs CH independent dependent chains for `iters` steps; one workgroup of 256 * W threads per CU.
// Reported: cycles per wave-instruction per SIMD, and for the packed kinds also cycles per scalar-equivalent op.
//   hipcc --offload-arch=gfx950 -O3 -o fp32_issue tools/ubench/fp32_issue.hip && ./fp32_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CH, int KIND>
__global__ void chains(float* out, int iters, float a, float b)
{
    float x[CH]; f2 v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { x[c] = a + c + threadIdx.x * 1e-6f; v[c] = f2{x[c], x[c] + 0.5f}; }
    const f2 vb = {b, b * 0.999f}, va = {a, a * 1.001f};
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (KIND == 0) x[c] = fmaf(x[c], b, a);                          // v_fma_f32
            else if (KIND == 1) v[c] = __builtin_elementwise_fma(v[c], vb, va);   // v_pk_fma_f32
            else if (KIND == 2) x[c] = x[c] * b;                             // v_mul_f32
            else if (KIND == 3) v[c] = v[c] * vb;                            // v_pk_mul_f32
            else if (KIND == 4) x[c] = x[c] + b;                             // v_add_f32
            else if (KIND == 5) v[c] = v[c] + vb;                            // v_pk_add_f32
            else if (KIND == 6) x[c] = __builtin_amdgcn_rcpf(x[c]) + a;      // v_rcp_f32 + add
            else if (KIND == 7) x[c] = __builtin_amdgcn_exp2f(x[c]) * b;     // v_exp_f32 + mul
            else if (KIND == 8) x[c] = fmaxf(x[c] * b, a);                   // mul + max
            else if (KIND == 9) x[c] = x[c] > a ? x[c] * b : x[c] + b;       // cmp + cndmask + mul + add
            else if (KIND == 10) { v[c].x = fmaxf(v[c].x, a); v[c].y = fmaxf(v[c].y, a); v[c] = v[c] * vb; }   // 2 max + pk_mul
        }
      }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c] + v[c].x + v[c].y;
    if (s == 12345.678f) out[0] = s;
}
template <int CH, int KIND>
static float run(int waves_per_simd, int iters, float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256), block(256 * waves_per_simd);       // one workgroup per CU: waves_per_simd waves on each SIMD
    chains<CH, KIND><<<grid, block>>>(d, 16, 1.0f, 0.999999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chains<CH, KIND><<<grid, block>>>(d, iters, 1.0f, 0.999999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float* d; hipMalloc(&d, 8);
    const int iters = 64000;
    const char* names[] = {"fma", "pk_fma", "mul", "pk_mul", "add", "pk_add", "rcp+add", "exp+mul", "mul+max", "cmp+cnd+mul+add", "2max+pk_mul"};
    const int ops[] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 4, 3};
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk);
#define ROW(K) for (int w = 1; w <= 4; ++w) { \
        float t1 = run<1, K>(w, iters, d), t2 = run<2, K>(w, iters, d), t4 = run<4, K>(w, iters, d); \
        double cyc = (double)clk * 1e3; \
        printf("%-16s waves/SIMD %d: cycles per wave-instr (per SIMD)  ILP1 %.2f  ILP2 %.2f  ILP4 %.2f\n", names[K], w, \
               t1 * 1e-3 * cyc / ((double)iters * 1 * ops[K] * w), t2 * 1e-3 * cyc / ((double)iters * 2 * ops[K] * w), \
               t4 * 1e-3 * cyc / ((double)iters * 4 * ops[K] * w)); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10)
    return 0;
}
