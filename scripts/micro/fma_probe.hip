// fp64 VALU throughput probe for gfx950: independent FMA chains per lane, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o fma_probe fma_probe.hip && ./fma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ __launch_bounds__(256) void k_fma(double *out, double a, double b, int iters)
{
    double v[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = __builtin_fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void k_addmax(double *out, double a, double b, int iters)
{
    double v[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) v[i] = fmax(v[i] + a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    double *out;
    const int blocks = 256 * 16, iters = 4096;
    hipMalloc(&out, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, double ops_per_thread) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double ops = ops_per_thread * blocks * 256.0;
        printf("%-28s %.3f ms  %.1f Gop/s (instr-lanes)  = %.1f TFLOP/s if FMA\n", name, ms, ops / ms / 1e6, 2 * ops / ms / 1e9);
    };
    run("fma 16 chains", k_fma<16>, 16.0 * iters);
    run("fma 8 chains", k_fma<8>, 8.0 * iters);
    run("fma 4 chains", k_fma<4>, 4.0 * iters);
    run("add+max 8 chains (2 ops)", k_addmax<8>, 16.0 * iters);
    return 0;
}
