// fp64 matrix-core throughput probe for gfx950: independent v_mfma_f64_16x16x4_f64 chains per wave, operands in
// registers, no memory traffic.   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k_mfma(double *out, double a0, double b0, int iters)
{
    d4 c[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) c[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    double *out;
    const int iters = 2048;
    hipMalloc(&out, 256 * 64 * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, int chains, int blocks) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double mfmas = (double)chains * iters * blocks * 4.0;        // 4 waves per block
        printf("%-40s %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, ms,
               mfmas * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
    };
    run("mfma f64 16x16x4, 4 chains, 1 wave/SIMD", k_mfma<4>, 4, 256);
    run("mfma f64 16x16x4, 8 chains, 1 wave/SIMD", k_mfma<8>, 8, 256);
    run("mfma f64 16x16x4, 16 chains, 1 wave/SIMD", k_mfma<16>, 16, 256);
    run("mfma f64 16x16x4, 4 chains, 4 waves/SIMD", k_mfma<4>, 4, 1024);
    run("mfma f64 16x16x4, 8 chains, 2 waves/SIMD", k_mfma<8>, 8, 512);
    return 0;
}
