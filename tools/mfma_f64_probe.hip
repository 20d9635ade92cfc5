// Back-to-back v_mfma_f64_16x16x4_f64 on independent accumulators, every SIMD busy: what the f64 matrix pipe sustains.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o /tmp/mfma_f64_probe && /tmp/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu)
{
    const int cus = 256, iters = 2000;
    double *out;
    hipMalloc(&out, (size_t)cus * wgs_per_cu * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<cus * wgs_per_cu, 256>>>(out, 10, 1.0, 2.0);
    hipEventRecord(e0);
    k<NACC><<<cus * wgs_per_cu, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)cus * wgs_per_cu * 4 * iters * NACC * 2048.0;
    printf("acc %d, %d waves/SIMD: %.3f ms, %.1f TFLOP/s\n", NACC, wgs_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<4>(1); run<4>(2); run<8>(1); run<2>(4);
    return 0;
}
