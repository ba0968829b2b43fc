// Instruction-fetch probe (tuning aid): the same dependent fp64 FMA work as a small loop body (fits the
// instruction buffer / a few cache lines) and as a long straight-line body (UNROLL x CH 8-byte VOP3 instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int UNROLL>
__global__ void body(double* out, int iters, double a)
{
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = a + c + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = fma(x[c], x[(c + 1) % CH], 0.5);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    if (s == 12345.678) out[0] = s;
}
template <int CH, int UNROLL>
static float run(int waves, int iters, double* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    body<CH, UNROLL><<<256, 256 * waves>>>(d, UNROLL, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    body<CH, UNROLL><<<256, 256 * waves>>>(d, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    double* d; hipMalloc(&d, 8);
    const int iters = 65536;
    const double cyc = 2.4e9;
#define ROW(CH, U) for (int w = 1; w <= 4; ++w) { float t = run<CH, U>(w, iters, d); \
        printf("ILP%d body %5d instr (%6d B)  waves/SIMD %d: %.2f cycles per wave-instr per SIMD\n", CH, CH * U, CH * U * 8, w, \
               t * 1e-3 * cyc / ((double)iters * CH * w)); }
    ROW(1, 16) ROW(1, 1024) ROW(1, 4096) ROW(2, 16) ROW(2, 2048)
    return 0;
}
