// v_mfma_f32_32x32x2_f32 (16 passes: 64 clocks) on gfx950: ONE dependent accumulator chain per wave with NV independent VALU
// instructions written behind every MFMA -- does a wave's own vector work ride in the shadow of its matrix chain, and how much of it?
// Three VALU flavours: v_fma_f32, v_med3_f32, and the sorted-insert slot of csrc/knn.hip (v_cmp -> SGPR mask, two v_cndmask, v_med3).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_mix.hip -o build_tmp/mfma_f32_mix && build_tmp/mfma_f32_mix
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k_mix(const unsigned* in, float* out, unsigned long long* clk, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float a = __uint_as_float(in[lane]), b = __uint_as_float(in[lane + 300]);
    float v[8];
    int w[8];
    for (int c = 0; c < 8; ++c) { v[c] = __uint_as_float(in[lane + 64 * c]); w[c] = in[lane + 64 * c + 7]; }
    const float x = __uint_as_float(in[lane + 700]), y = __uint_as_float(in[lane + 800]);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int r = (q * NV + n) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(x), "v"(y));
                if (KIND == 1) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(x), "v"(y));
                if (KIND == 2) {               // one insert slot = 4 instructions (counts as 4 of the NV)
                    if ((n & 3) == 0)
                        asm volatile("v_cmp_lt_f32 s[20:21], %2, %3\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %1, %1, %5, s[20:21]\n"
                                     "v_med3_f32 %0, %0, %2, %3"
                                     : "+v"(v[r]), "+v"(w[r]) : "v"(x), "v"(v[(r + 1) & 7]), "v"(w[(r + 1) & 7]), "v"(w[(r + 2) & 7]) : "s20", "s21");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    for (int c = 0; c < 8; ++c) sum += v[c] + (float)w[c];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NV, int KIND>
void run_mix(int wps, const unsigned* in, float* out, unsigned long long* clk) {
    const int iters = 1000, blocks = 256 * wps;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<NV, KIND>), dim3(blocks), dim3(256), 0, 0, in, out, clk, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<NV, KIND>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * 32;
    static const char* names[] = {"v_fma_f32", "v_med3_f32", "insert slot (cmp, 2 cndmask, med3)"};
    printf("f32 chain + %2d x %-36s waves/SIMD %d: %7.1f ns per MFMA and SIMD   wave clocks per MFMA %.1f\n", NV, names[KIND], wps,
           ms * 1e6 / (mf * wps), (double)c / mf);
}

int main() {
    unsigned* in; float* out; unsigned long long* clk;
    (void)hipMalloc(&in, 1 << 20); (void)hipMalloc(&out, 256 * 256 * 8 * 4); (void)hipMalloc(&clk, 8);
    (void)hipMemset(in, 0x3c, 1 << 20);
    for (int w = 1; w <= 2; ++w) {
        run_mix<0, 0>(w, in, out, clk);
        run_mix<4, 0>(w, in, out, clk);
        run_mix<8, 0>(w, in, out, clk);
        run_mix<12, 0>(w, in, out, clk);
        run_mix<16, 0>(w, in, out, clk);
        run_mix<8, 1>(w, in, out, clk);
        run_mix<12, 1>(w, in, out, clk);
        run_mix<16, 1>(w, in, out, clk);
        run_mix<8, 2>(w, in, out, clk);
        run_mix<12, 2>(w, in, out, clk);
        run_mix<16, 2>(w, in, out, clk);
    }
    return 0;
}
