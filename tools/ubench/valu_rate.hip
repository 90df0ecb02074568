// VALU issue rates on gfx950, whole chip: independent chains of one opcode per wave, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o build_tmp/valu_rate && build_tmp/valu_rate
// Prints wave-instructions per clock per SIMD (at the event-timed wall clock and an assumed 2.4 GHz) for
// the opcodes of the selection / gather kernels (k_cmp_exec3mov / k_cmp_3cnd: clocks per GROUP -- a compare plus three conditional updates, by exec-masked moves or by v_cndmask).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define NCHAIN 16
#define BODY_REPS 8

#define KERNEL(NAME, ASM)                                                                          \
    template <int WPS>                                                                             \
    __global__ __launch_bounds__(256, WPS) void NAME(const float* in, float* out, int iters) {     \
        float v[NCHAIN];                                                                           \
        for (int c = 0; c < NCHAIN; ++c) v[c] = in[threadIdx.x + 64 * c];                          \
        const float a = in[threadIdx.x + 1024], b = in[threadIdx.x + 1100];                        \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int r = 0; r < BODY_REPS; ++r) {                                \
                _Pragma("unroll") for (int c = 0; c < NCHAIN; ++c) { ASM }                         \
            }                                                                                      \
        }                                                                                          \
        float s = 0.f;                                                                             \
        for (int c = 0; c < NCHAIN; ++c) s += v[c];                                                \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                   \
    }

KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));)
KERNEL(k_fmac, asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));)
KERNEL(k_mul, asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_max, asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_cnd_vcc, asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(a) : );)
KERNEL(k_cnd_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[c]) : "v"(a) : );)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(a) : "vcc");)
KERNEL(k_max_i32, asm volatile("v_max_i32_e32 %0, %0, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_max_i32_c0, asm volatile("v_max_i32_e32 %0, 0, %0" : "+v"(v[c]));)
KERNEL(k_max_f32_c0, asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(v[c]));)
KERNEL(k_min_f32, asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_mov, asm volatile("v_mov_b32_e32 %0, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_mov_s, asm volatile("v_mov_b32_e32 %0, s20" : "+v"(v[c]));)
KERNEL(k_add_u32, asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_and, asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_med3, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));)
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));)
KERNEL(k_cmp, asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0" : "+v"(v[c]) : "v"(a) : "vcc");)
KERNEL(k_cmp_u32, asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0" : "+v"(v[c]) : "v"(a) : "vcc");)
KERNEL(k_fmac_s, asm volatile("v_fmac_f32_e32 %0, s20, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_mul_s, asm volatile("v_mul_f32_e32 %0, s20, %0" : "+v"(v[c]));)
KERNEL(k_cmp_exec3mov, asm volatile("v_cmp_gt_f32_e32 vcc, %3, %0\n\ts_and_saveexec_b64 s[22:23], vcc\n\tv_mov_b32_e32 %0, %3\n\tv_mov_b32_e32 %1, %4\n\tv_mov_b32_e32 %2, s20\n\ts_mov_b64 exec, s[22:23]" : "+v"(v[c]), "+v"(v[(c + 5) & 15]), "+v"(v[(c + 9) & 15]) : "v"(a), "v"(b) : "vcc", "scc", "s22", "s23");)
KERNEL(k_cmp_3cnd, asm volatile("v_cmp_gt_f32_e32 vcc, %3, %0\n\tv_cndmask_b32_e32 %0, %0, %3, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc\n\tv_cndmask_b32_e32 %2, %2, %3, vcc" : "+v"(v[c]), "+v"(v[(c + 5) & 15]), "+v"(v[(c + 9) & 15]) : "v"(a), "v"(b) : "vcc");)
KERNEL(k_cmpx_3mov, asm volatile("s_mov_b64 s[22:23], exec\n\tv_cmpx_gt_f32_e32 vcc, %3, %0\n\tv_mov_b32_e32 %0, %3\n\tv_mov_b32_e32 %1, %4\n\tv_mov_b32_e32 %2, %4\n\ts_mov_b64 exec, s[22:23]" : "+v"(v[c]), "+v"(v[(c + 5) & 15]), "+v"(v[(c + 9) & 15]) : "v"(a), "v"(b) : "vcc", "s22", "s23");)
KERNEL(k_fma_clamp, asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(v[c]) : "v"(a), "v"(b));)
KERNEL(k_max_mul, asm volatile("v_max_f32_e32 %0, 0, %0\n\tv_mul_f32_e32 %0, %0, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 5, %1" : "+v"(v[c]) : "v"(a));)
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[c]) : "v"(a), "v"(b));)
KERNEL(k_ashr, asm volatile("v_ashrrev_i32_e32 %0, 31, %0" : "+v"(v[c]));)
KERNEL(k_cmp_sgpr_cnd, asm volatile("v_cmp_gt_f32_e64 s[20:21], %1, %0\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[c]) : "v"(a) : "s20", "s21");)

template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_pkfma(const float* in, float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v[NCHAIN];
    for (int c = 0; c < NCHAIN; ++c) { v[c].x = in[threadIdx.x + 64 * c]; v[c].y = in[threadIdx.x + 64 * c + 7]; }
    f2 a, b; a.x = in[threadIdx.x + 1024]; a.y = in[threadIdx.x + 1030]; b.x = in[threadIdx.x + 1100]; b.y = in[threadIdx.x + 1111];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < BODY_REPS; ++r) {
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
    for (int c = 0; c < NCHAIN; ++c) s += v[c].x + v[c].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_pkmul(const float* in, float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v[NCHAIN];
    for (int c = 0; c < NCHAIN; ++c) { v[c].x = in[threadIdx.x + 64 * c]; v[c].y = in[threadIdx.x + 64 * c + 7]; }
    f2 a; a.x = in[threadIdx.x + 1024]; a.y = in[threadIdx.x + 1030];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < BODY_REPS; ++r) {
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(a));
        }
    }
    float s = 0.f;
    for (int c = 0; c < NCHAIN; ++c) s += v[c].x + v[c].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// ds_read_b128, 16 independent reads per wait
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_dsr128(const float* in, float* out, int iters) {
    __shared__ float4 sm[1024];
    for (int t = threadIdx.x; t < 1024; t += 256) sm[t] = make_float4(in[t], in[t + 1], in[t + 2], in[t + 3]);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    const int base = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < BODY_REPS; ++r) {
            float4 t[NCHAIN];
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) t[c] = sm[(base + 64 * c + it) & 1023];
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) { acc.x += t[c].x; acc.y += t[c].w; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}

template <typename K>
static void run(const char* name, K kern, int wps, double inst_per_iter, const float* in, float* out) {
    const int blocks = 256 * wps, iters = 2000;
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
    // wave-instructions per SIMD: each SIMD hosts wps waves, each issuing inst_per_iter * iters
    const double winst = inst_per_iter * iters * wps;
    const double clk = best * 1e-3 * 2.4e9;
    setvbuf(stdout, NULL, _IONBF, 0); printf("%-28s wps %d: %8.1f us  %.2f clk per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, best * 1e3, clk / winst);
}

int main() {
    std::vector<float> h(4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + 0.001f * (float)(i % 97);
    float *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const double n = (double)NCHAIN * BODY_REPS;
#define RUN(K, MULT) run(#K, K<2>, 2, n * MULT, in, out); run(#K, K<4>, 4, n * MULT, in, out);
    RUN(k_fma, 1) RUN(k_fmac, 1) RUN(k_mul, 1) RUN(k_max, 1) RUN(k_pkfma, 1) RUN(k_pkmul, 1)
    RUN(k_cnd_sgpr, 1) RUN(k_cmp_cnd, 2) RUN(k_cmp_sgpr_cnd, 2)
    RUN(k_max_i32, 1) RUN(k_max_i32_c0, 1) RUN(k_max_f32_c0, 1) RUN(k_min_f32, 1) RUN(k_mov, 1) RUN(k_mov_s, 1) RUN(k_add_u32, 1) RUN(k_and, 1)
    RUN(k_med3, 1) RUN(k_max3, 1) RUN(k_cmp, 1) RUN(k_cmp_u32, 1) RUN(k_fmac_s, 1) RUN(k_mul_s, 1)
    RUN(k_cmp_exec3mov, 1) RUN(k_cmp_3cnd, 1) RUN(k_cmpx_3mov, 1) RUN(k_fma_clamp, 1) RUN(k_max_mul, 2) RUN(k_lshl_or, 1) RUN(k_bfi, 1) RUN(k_ashr, 1)
    return 0;
}
