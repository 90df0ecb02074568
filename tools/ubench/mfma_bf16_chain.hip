// v_mfma_f32_32x32x16_bf16 on gfx950: issue rate of ONE dependent accumulator chain per wave (the panel form of csrc/gemm_x3.hip),
// of two / four independent chains, and of the panel kernel's kstep (3 ds_read_b128 + 6 MFMAs on one accumulator), at 1, 2 and 3
// waves per SIMD; the shader clock the chip actually holds under that load (s_memtime against the event clock).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_chain.hip -o build_tmp/mfma_bf16_chain && build_tmp/mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int NCH, bool LDS>
__global__ __launch_bounds__(256) void k_chain(const unsigned* in, float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) char sm[3 * 32 * 256];
    for (int i = threadIdx.x; i < 3 * 32 * 256 / 4; i += 256) reinterpret_cast<unsigned*>(sm)[i] = in[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    u32x4 a[3], b[3];
    for (int p = 0; p < 3; ++p)
        for (int e = 0; e < 4; ++e) { a[p][e] = in[lane + 64 * p + e]; b[p][e] = in[lane + 300 + 64 * p + e]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if constexpr (LDS) {
                const int off = li * 256 + (((2 * s + lh) ^ (li & 15)) << 4);
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const u32x4*>(sm + p * 8192 + off);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q)
                acc[q % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q % 3]), __builtin_bit_cast(bf16x8, b[(q + s) % 3]),
                                                                        acc[q % NCH], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int c = 0; c < NCH; ++c)
        for (int r = 0; r < 16; ++r) sum += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

// one dependent MFMA chain with NV independent VALU instructions (v_fma_f32 on 8 rotating registers) after every MFMA: do the vector
// ALU and the matrix pipe of a SIMD run side by side?
template <int NV>
__global__ __launch_bounds__(256) void k_mix(const unsigned* in, float* out, unsigned long long* clk, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 a, b;
    for (int e = 0; e < 4; ++e) { a[e] = in[lane + e]; b[e] = in[lane + 300 + e]; }
    float v[8];
    for (int c = 0; c < 8; ++c) v[c] = __uint_as_float(in[lane + 64 * c]);
    const float x = __uint_as_float(in[lane + 700]), y = __uint_as_float(in[lane + 800]);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 48; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(q * NV + n) & 7]) : "v"(x), "v"(y));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc[r];
    for (int c = 0; c < 8; ++c) sum += v[c];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NV>
void run_mix(int wps, const unsigned* in, float* out, unsigned long long* clk) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<NV>), dim3(blocks), dim3(256), 0, 0, in, out, clk, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<NV>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * 48;
    printf("chain + %d v_fma per MFMA        waves/SIMD %d: %8.3f ms  %7.1f ns per MFMA and SIMD   wave clocks per MFMA %.1f\n", NV, wps, ms,
           ms * 1e6 / (mf * wps), (double)c / mf);
}

template <int NCH, bool LDS>
void run(const char* name, int wps, const unsigned* in, float* out, unsigned long long* clk) {
    const int iters = 2000;
    const int blocks = 256 * wps;            // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<NCH, LDS>), dim3(blocks), dim3(256), 0, 0, in, out, clk, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain<NCH, LDS>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * 48;                        // MFMAs per wave
    const double tflops = (double)blocks * 4 * mf * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD %d: %8.3f ms  %7.1f ns per MFMA and SIMD  %7.1f TFLOP/s  s_memtime ticks %llu (%.1f MHz counter)\n", name, wps, ms,
           ms * 1e6 / (mf * wps), tflops, c, (double)c / (ms * 1e3));
}

int main() {
    unsigned* in; float* out; unsigned long long* clk;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 256 * 256 * 8 * 4); hipMalloc(&clk, 8);
    hipMemset(in, 0x3c, 1 << 20);
    for (int w = 1; w <= 3; ++w) {
        run<1, false>("one chain", w, in, out, clk);
        run<2, false>("two chains", w, in, out, clk);
        run<3, false>("three chains", w, in, out, clk);
        run<1, true>("one chain + 3 ds_read_b128", w, in, out, clk);
        run<2, true>("two chains + 3 ds_read_b128", w, in, out, clk);
    }
    for (int w = 1; w <= 2; ++w) {
        run_mix<0>(w, in, out, clk);
        run_mix<2>(w, in, out, clk);
        run_mix<4>(w, in, out, clk);
        run_mix<6>(w, in, out, clk);
        run_mix<8>(w, in, out, clk);
        run_mix<12>(w, in, out, clk);
    }
    return 0;
}
