// Per-instruction cost of the fp64 VALU operations the integration loops use, as long straight-line dependent
// chains (no loop overhead): cycles per chain STEP, single wave and three waves per SIMD (tuning aid).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND, int UNROLL>
__global__ void body(double* out, int iters, double a, double b)
{
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int i = 0; i < iters; i += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (KIND == 0) x = fma(x, y, 0.5);
            else if (KIND == 1) x = __builtin_amdgcn_rcp(x) + y;                       // rcp + add
            else if (KIND == 2) x = rint(x * y);                                       // mul + rndne
            else if (KIND == 3) x = ldexp(x, (int)y) * y;                              // (cvt hoisted) ldexp + mul
            else if (KIND == 4) x = ldexp(y, (int)x) + x;                              // cvt_i32_f64 + ldexp + add
            else if (KIND == 5) x = (x >= y) ? x * y : y;                              // cmp + 2 cndmask + mul
            else if (KIND == 6) x = fmax(x * y, y);                                    // mul + max
            else if (KIND == 7) x = fmin(fmax(x * y, 0.25), 4.0);                      // mul + max + min
        }
    }
    if (x == 12345.678) out[0] = x;
}
template <int KIND>
static void run(const char* name, int steps_instr, double* d)
{
    const int iters = 32768;
    for (int w = 1; w <= 3; w += 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        body<KIND, 512><<<256, 256 * w>>>(d, 512, 1.0, 0.999);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        body<KIND, 512><<<256, 256 * w>>>(d, iters, 1.0, 0.999);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-26s (%d instr/step) waves/SIMD %d: %6.2f cycles per step per wave-slot (%.2f per instr)\n", name, steps_instr, w,
               ms * 1e-3 * 2.4e9 / ((double)iters * w), ms * 1e-3 * 2.4e9 / ((double)iters * w * steps_instr));
    }
}
int main()
{
    double* d; hipMalloc(&d, 8);
    run<0>("fma", 1, d); run<1>("rcp + add", 2, d); run<2>("mul + rndne", 2, d); run<3>("ldexp + mul", 2, d);
    run<4>("cvt_i32 + ldexp + add", 3, d); run<5>("cmp + 2 cndmask + mul", 4, d); run<6>("mul + max", 2, d);
    run<7>("mul + max + min", 3, d);
    return 0;
}
