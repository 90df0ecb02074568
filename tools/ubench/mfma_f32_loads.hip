// fp32 MFMA rate with operand loads in the loop: does buffer-load traffic (L1 hits) next to back-to-back MFMAs cost MFMA rate or clock?
//   per iteration: NL x 16-byte buffer loads (same 1 KB per wave each time -> L1 hits) + 16 MFMAs on 4 accumulators
//   prints TF and the effective shader clock (s_memtime cycles of one wave / wall time)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int NL, bool USE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float* in, float* out, int iters, long long* cyc) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 1 << 20, 0x00020000);
    const unsigned voff = (threadIdx.x & 63) * 16 + (blockIdx.x & 7) * 4096;
    u32x4 v[8];
    for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * 1024, 0);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u32x4 n[8];
#pragma unroll
        for (int j = 0; j < NL; ++j) n[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((it + j) & 31) * 1024, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(v[j][a]), __uint_as_float(v[4 + (j & 3)][a]), acc[a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (USE) {
#pragma unroll
            for (int j = 0; j < NL; ++j) v[j] = n[j];
        } else {
#pragma unroll
            for (int j = 0; j < NL; ++j) asm volatile("" ::"v"(n[j]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <typename K>
static void run(const char* name, K kern, int blocks, int iters, const float* in, float* out, long long* cyc) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, iters, cyc);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, iters, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double fl = 16.0 * 4096 * iters * blocks * 4.0;
    printf("%-40s blocks %4d: %8.1f us  %7.1f TF   memtime ticks %lld -> %.0f MHz-equivalent\n", name, blocks, best * 1e3, fl / best / 1e9, c, c / (best * 1e3));
}
int main() {
    std::vector<float> h(1 << 18);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *out; long long* cyc;
    (void)hipMalloc(&in, 1 << 20); (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&cyc, 8);
    (void)hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    const int it = 4000;
    for (int blocks : {512, 256}) {
        run("16 MFMA, no loads", k<0, false, 2>, blocks, it, in, out, cyc);
        run("16 MFMA + 5 loads (unused)", k<5, false, 2>, blocks, it, in, out, cyc);
        run("16 MFMA + 5 loads (operands)", k<5, true, 2>, blocks, it, in, out, cyc);
        run("16 MFMA + 8 loads (operands)", k<8, true, 2>, blocks, it, in, out, cyc);
        run("16 MFMA + 2 loads (operands)", k<2, true, 2>, blocks, it, in, out, cyc);
    }
    return 0;
}
