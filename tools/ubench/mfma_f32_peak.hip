// fp32 MFMA ceiling on the whole chip: N back-to-back v_mfma_f32_32x32x2_f32 / 16x16x4 on NA accumulators per wave, W waves per SIMD
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_peak.hip -o build_tmp/mfma_f32_peak && build_tmp/mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NA, int WPS>
__global__ __launch_bounds__(256, WPS) void k32(const float* in, float* out, int iters) {
    f32x16 acc[NA];
    for (int a = 0; a < NA; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av[4], bv[4];
    for (int j = 0; j < 4; ++j) { av[j] = in[threadIdx.x + 256 * j]; bv[j] = in[1024 + threadIdx.x + 256 * j]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[(j + a) & 3], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NA; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NA, int WPS>
__global__ __launch_bounds__(256, WPS) void k16(const float* in, float* out, int iters) {
    f32x4 acc[NA];
    for (int a = 0; a < NA; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    float av[4], bv[4];
    for (int j = 0; j < 4; ++j) { av[j] = in[threadIdx.x + 256 * j]; bv[j] = in[1024 + threadIdx.x + 256 * j]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[(j + a) & 3], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NA; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
static void run(const char* name, K kern, int blocks, int iters, double flop_per_iter_per_wave, const float* in, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double fl = flop_per_iter_per_wave * iters * blocks * 4.0;
    printf("%-34s blocks %4d: %8.1f us  %7.1f TF\n", name, blocks, best * 1e3, fl / best / 1e9);
}
int main(int argc, char** argv) {
    const bool zero = argc > 1 && argv[1][0] == 'z';
    std::vector<float> h(2048);
    srand(1);
    for (auto& v : h) v = zero ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *out; hipMalloc(&in, 2048 * 4); hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(in, h.data(), 2048 * 4, hipMemcpyHostToDevice);
    printf("data: %s\n", zero ? "zeros" : "uniform [-1,1)");
    const int it = 2000;
    for (int blocks : {256, 512, 128}) {
        run("32x32x2 NA=4 wps1 (launch bound)", k32<4, 1>, blocks, it, 4.0 * 4 * 4096, in, out);
        run("32x32x2 NA=4 wps2", k32<4, 2>, blocks, it, 4.0 * 4 * 4096, in, out);
        run("32x32x2 NA=8 wps1", k32<8, 1>, blocks, it, 4.0 * 8 * 4096, in, out);
        run("32x32x2 NA=1 wps2", k32<1, 2>, blocks, it, 4.0 * 1 * 4096, in, out);
        run("16x16x4 NA=4 wps2", k16<4, 2>, blocks, it, 4.0 * 4 * 2048, in, out);
        run("16x16x4 NA=16 wps2", k16<16, 2>, blocks, it, 4.0 * 16 * 2048, in, out);
    }
    return 0;
}
