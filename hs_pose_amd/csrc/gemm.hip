// gemm.hip -- split-K weight-gradient GEMM on the fp32 matrix cores of gfx950.
//
//   C[m][n] = sum_k A[k][m] * B[k][n]          A (K x M, row stride lda), B (K x N, row stride ldb)
//   colsum[n] = sum_k B[k][n]                  (optional: the bias gradient, fused)
//
// This is the parameter-gradient shape of every dense op of the HS stack (reference: autograd of
// `feature_map @ self.weights + self.bias`, network/fs_net_repo/gcn3d.py:171, and of the 1x1 Conv1d
// layers :85,:149,:186): K = B*N point rows (16 448 at B=16, N=1028) against a small M x N weight
// (128 x 1024 ...).  A library GEMM sees 32 output tiles and no parallelism over K (measured 36 TF/s);
// here K is split over up to ~2000 independent waves.
//
// Each wave owns a 64x64 output tile for one K slice: 4 accumulators of v_mfma_f32_32x32x2_f32
// (fp32 in, fp32 accumulate, an exact k-ordered fma chain).  Both operands are k-major rows, which
// is exactly the MFMA operand layout (lane l supplies A[k = 2s + (l>>5)][m], B[k][n]): a lane reads one
// float2 of A and one of B per k-pair straight from global memory -- the two components feed the
// even-row / odd-row (even-col / odd-col) MFMA tiles, so every half-wave reads 256 contiguous bytes;
// no LDS, no barriers, register double-buffered prefetch 4 k-pairs deep.  Partials go to a workspace
// and are folded in a fixed order (deterministic) by wgrad_reduce_kernel, which writes C with an
// arbitrary leading dimension (so a gradient can land in a column block of a larger tensor).
#include "common.h"

namespace hsp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#define WG_UNROLL 4   // k-pairs per prefetch group

// KB = 1: the 4 waves of a workgroup take 4 neighbouring tiles of one K slice (they share B rows in L1).
// KB = 4: they take 4 consecutive K slices of ONE tile and fold their accumulators through LDS (fixed order), so
//         a small output with a long K (128x128 <- 16448 rows) can be cut into hundreds of slices -- enough waves to
//         fill the chip -- without multiplying the partial sums the reduce kernel has to read.
// FT: storage type of A and B (float, or bf16_t: the pair of columns a lane owns is one 4-byte load, widened to fp32 --
// the products still run on the fp32 MFMA, exact and in the same order as for fp32 operands)
template <bool COLSUM, int KB, typename FT>
__global__ __launch_bounds__(256) void wgrad_kernel(const FT* __restrict__ A, int lda,
                                                    const FT* __restrict__ B, int ldb, int M, int N, int K,
                                                    int SK, int kslice, float* __restrict__ part,
                                                    float* __restrict__ cs_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and provably so
    const int tiles = (M >> 6) * (N >> 6);
    int slice, t;
    if (KB == 1) {
        const int w = blockIdx.x * 4 + wv;
        if (w >= tiles * SK) return;
        slice = w / tiles; t = w - slice * tiles;
    } else {
        const int grp = blockIdx.x / tiles;
        t = blockIdx.x - grp * tiles;
        slice = grp * 4 + wv;                               // slices past K read zeros
    }
    const int tiles_m = M >> 6;
    const int tn = t / tiles_m, tm = t - tn * tiles_m;   // tm fastest: the waves of a block share B rows (L1/L2 reuse)
    const int m0 = tm << 6, n0 = tn << 6;
    const int k0 = slice * kslice;
    const int k1 = min(K, k0 + kslice);
    const int h = lane >> 5, i = lane & 31;
    // operands by raw buffer loads: descriptor over the whole matrix, per-lane byte offset fixed for the kernel
    // (row parity h, columns 2i, 2i+1 of the tile), row pair selected by a scalar offset -> no vector address
    // arithmetic in the loop, and rows >= K (ragged end of the last slice; slices are multiples of 8 rows) read 0
    constexpr int ES = sizeof(FT);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<FT*>(A), 0, (int)((((size_t)K - 1) * lda + M) * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<FT*>(B), 0, (int)((((size_t)K - 1) * ldb + N) * ES), 0x00020000);
    const int va = (h * lda + m0 + 2 * i) * ES, vb = (h * ldb + n0 + 2 * i) * ES;
    f32x16 c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
    float2 cs = make_float2(0.f, 0.f);

    // two register buffers, used alternately (no copies): the loads of group g+1 are in flight under the 16 MFMAs
    // of group g and are first touched when group g+1 starts
    float2 a0[WG_UNROLL], b0[WG_UNROLL], a1[WG_UNROLL], b1[WG_UNROLL];
    auto load_group = [&](int kk, float2 (&aa)[WG_UNROLL], float2 (&bb)[WG_UNROLL]) {
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
            if constexpr (ES == 4) {
                const auto x = __builtin_amdgcn_raw_buffer_load_b64(ra, va, (kk + 2 * u) * lda * 4, 0);
                const auto y = __builtin_amdgcn_raw_buffer_load_b64(rb, vb, (kk + 2 * u) * ldb * 4, 0);
                aa[u] = make_float2(__int_as_float(x[0]), __int_as_float(x[1]));
                bb[u] = make_float2(__int_as_float(y[0]), __int_as_float(y[1]));
            } else {
                const unsigned x = __builtin_amdgcn_raw_buffer_load_b32(ra, va, (kk + 2 * u) * lda * 2, 0);
                const unsigned y = __builtin_amdgcn_raw_buffer_load_b32(rb, vb, (kk + 2 * u) * ldb * 2, 0);
                aa[u] = make_float2(__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u));
                bb[u] = make_float2(__uint_as_float(y << 16), __uint_as_float(y & 0xffff0000u));
            }
        }
    };
    auto mma_group = [&](const float2 (&aa)[WG_UNROLL], const float2 (&bb)[WG_UNROLL]) {
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
            c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u].x, bb[u].x, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u].x, bb[u].y, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u].y, bb[u].x, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u].y, bb[u].y, c11, 0, 0, 0);
            if (COLSUM) { cs.x += bb[u].x; cs.y += bb[u].y; }
        }
    };
    constexpr int GR = 2 * WG_UNROLL;                       // k rows per group
    // the next group is requested unconditionally (past the slice it is simply not used; past K it reads 0):
    // a branch around the loads would make the compiler wait for them at the join
    // Branch-free body (slices are multiples of 2 groups; the ragged end of the last slice multiplies zeros): with
    // an early exit between a load group and its MFMAs hipcc sinks the loads behind the branch, next to their
    // use, and nothing is prefetched.  sched_barrier keeps the machine scheduler from interleaving them back.
    load_group(k0, a0, b0);
    for (int kk = k0; kk < k1; kk += 2 * GR) {
        load_group(kk + GR, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_group(kk + 2 * GR, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // accumulator r <-> tile row (r&3) + 8*(r>>2) + 4*h, tile col i; tile (ta,tb) holds C[m0+2*row+ta][n0+2*i+tb]
    if (KB == 1) {
        float* pc = part + (size_t)slice * M * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            *reinterpret_cast<float2*>(pc + (size_t)(m0 + 2 * row) * N + n0 + 2 * i) = make_float2(c00[r], c01[r]);
            *reinterpret_cast<float2*>(pc + (size_t)(m0 + 2 * row + 1) * N + n0 + 2 * i) = make_float2(c10[r], c11[r]);
        }
        if (COLSUM && tm == 0) {
            cs.x += __shfl_xor(cs.x, 32);
            cs.y += __shfl_xor(cs.y, 32);
            if (h == 0) *reinterpret_cast<float2*>(cs_part + (size_t)slice * N + n0 + 2 * i) = cs;
        }
        return;
    }
    // KB == 4: buf[wave][acc][r][lane]; wave a then folds accumulator a of the four waves (order 0..3) and stores its
    // interleaved quarter of the tile; the column sums go through csb[wave][lane]
    float* buf = reinterpret_cast<float*>(smem);
    float2* csb = reinterpret_cast<float2*>(buf + 4 * 4 * 16 * 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        buf[((wv * 4 + 0) * 16 + r) * 64 + lane] = c00[r];
        buf[((wv * 4 + 1) * 16 + r) * 64 + lane] = c01[r];
        buf[((wv * 4 + 2) * 16 + r) * 64 + lane] = c10[r];
        buf[((wv * 4 + 3) * 16 + r) * 64 + lane] = c11[r];
    }
    if (COLSUM) csb[wv * 64 + lane] = cs;
    __syncthreads();
    const int grp = blockIdx.x / tiles;
    float* pc = part + (size_t)grp * M * N;
    const int ta = wv >> 1, tb = wv & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = buf[((0 * 4 + wv) * 16 + r) * 64 + lane];
        v += buf[((1 * 4 + wv) * 16 + r) * 64 + lane];
        v += buf[((2 * 4 + wv) * 16 + r) * 64 + lane];
        v += buf[((3 * 4 + wv) * 16 + r) * 64 + lane];
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        pc[(size_t)(m0 + 2 * row + ta) * N + n0 + 2 * i + tb] = v;
    }
    if (COLSUM && tm == 0 && wv == 0) {
        float2 c = csb[lane];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) { c.x += csb[w2 * 64 + lane].x; c.y += csb[w2 * 64 + lane].y; }
        c.x += __shfl_xor(c.x, 32);
        c.y += __shfl_xor(c.y, 32);
        if (h == 0) *reinterpret_cast<float2*>(cs_part + (size_t)grp * N + n0 + 2 * i) = c;
    }
}

// C[m][n] = sum_s part[s][m][n]; likewise colsum.  Workgroup = 64 float4 elements x 4 slice groups:
// group g sums slices g, g+4, ... with loads in flight, the 4 groups are folded through LDS in a
// fixed order (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int SK, int M, int N,
                                                           float* __restrict__ C, int ldc,
                                                           const float* __restrict__ cs_part,
                                                           float* __restrict__ colsum) {
    __shared__ float4 red[4][64];
    const int le = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int nq = N >> 2;
    const long long total = (long long)M * nq;                 // float4 elements of C
    const long long ncs = colsum ? (N >> 2) : 0;               // float4 elements of colsum
    const long long e = (long long)blockIdx.x * 64 + le;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = nullptr;
    size_t stride = 0;
    if (e < total) { src = part + (size_t)e * 4; stride = (size_t)M * N; }
    else if (e < total + ncs) { src = cs_part + (size_t)(e - total) * 4; stride = (size_t)N; }
    if (src) {
        float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f);
        int sl = sg;
        for (; sl + 4 < SK; sl += 8) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)sl * stride);
            const float4 u = *reinterpret_cast<const float4*>(src + (size_t)(sl + 4) * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            s2.x += u.x; s2.y += u.y; s2.z += u.z; s2.w += u.w;
        }
        if (sl < SK) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)sl * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        s.x += s2.x; s.y += s2.y; s.z += s2.z; s.w += s2.w;
    }
    red[sg][le] = s;
    __syncthreads();
    if (sg == 0 && src) {
        float4 r = red[0][le];
#pragma unroll
        for (int g = 1; g < 4; ++g) { const float4 v = red[g][le]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        if (e < total) {
            const int m = (int)(e / nq), q = (int)(e - (long long)m * nq);
            float* c = C + (size_t)m * ldc + (q << 2);
            c[0] = r.x; c[1] = r.y; c[2] = r.z; c[3] = r.w;    // ldc need not be a multiple of 4
        } else {
            *reinterpret_cast<float4*>(colsum + (size_t)(e - total) * 4) = r;
        }
    }
}

// (slices, rows per slice, waves of a workgroup along K)
static int wgrad_pick_sk(int M, int N, int K, int* kslice, int* kblock) {
    const int tiles = (M >> 6) * (N >> 6);
    constexpr int G = 4 * WG_UNROLL;                           // whole pairs of prefetch groups
    if (tiles <= 64) {
        // few tiles: 4 K slices per workgroup, folded in LDS; >= 32 rows per slice, <= 64 partial sums
        int sk = 2048 / tiles;
        if (sk > K / 32) sk = K / 32;
        if (sk > 256) sk = 256;
        if (sk >= 8) {
            int ks = (K + sk - 1) / sk;
            ks = (ks + G - 1) / G * G;
            sk = (K + ks - 1) / ks;
            sk = (sk + 3) & ~3;
            *kslice = ks;
            *kblock = 4;
            return sk;
        }
    }
    int sk = (2048 + tiles - 1) / tiles;                       // ~2 waves per SIMD
    int max_sk = (K + 127) / 128;                              // at least 128 rows per slice
    if (max_sk > 64) max_sk = 64;                              // bound the partial-sum traffic
    if (sk > max_sk) sk = max_sk;
    if (sk < 1) sk = 1;
    int ks = (K + sk - 1) / sk;
    ks = (ks + G - 1) / G * G;
    sk = (K + ks - 1) / ks;
    *kslice = ks;
    *kblock = 1;
    return sk;
}

}  // namespace hsp

using namespace hsp;

extern "C" size_t hsp_wgrad_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int ks, kb;
    const int sk = wgrad_pick_sk(M, N, K, &ks, &kb);
    return (size_t)(sk / kb) * ((size_t)M * N + N) * sizeof(float);
}

template <bool COLSUM, typename FT>
static int wgrad_launch(const FT* A, int lda, const FT* B, int ldb, int M, int N, int K, int sk, int ks, int kb,
                        float* part, float* cs_part, hipStream_t st) {
    const int tiles = (M >> 6) * (N >> 6);
    if (kb == 1) {
        hipLaunchKernelGGL((wgrad_kernel<COLSUM, 1, FT>), dim3((tiles * sk + 3) / 4), dim3(256), 0, st, A, lda, B, ldb, M, N, K,
                           sk, ks, part, cs_part);
        return check_launch();
    }
    const size_t lds = (size_t)4 * 4 * 16 * 64 * sizeof(float) + (size_t)4 * 64 * sizeof(float2);
    auto kern = wgrad_kernel<COLSUM, 4, FT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    hipLaunchKernelGGL(kern, dim3(tiles * (sk / 4)), dim3(256), lds, st, A, lda, B, ldb, M, N, K, sk, ks, part, cs_part);
    return check_launch();
}

template <typename FT>
static int wgrad_impl(const FT* A, int lda, const FT* B, int ldb, int M, int N, int K, float* C, int ldc,
                      float* colsum_B, void* ws, size_t ws_bytes, hspStream_t stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda < M || ldb < N || ldc < N) return HSP_ERR_BAD_ARG;
    if ((M & 63) || (N & 63) || (lda & 1) || (ldb & 1)) return HSP_ERR_UNSUPPORTED;   // 64x64 wave tiles, float2 loads
    if (!ws || ws_bytes < hsp_wgrad_workspace_bytes(M, N, K)) return HSP_ERR_WORKSPACE;
    int ks, kb;
    const int sk = wgrad_pick_sk(M, N, K, &ks, &kb);
    const int nparts = sk / kb;
    float* part = reinterpret_cast<float*>(ws);
    float* cs_part = part + (size_t)nparts * M * N;
    hipStream_t st = as_stream(stream);
    int rc = colsum_B ? wgrad_launch<true, FT>(A, lda, B, ldb, M, N, K, sk, ks, kb, part, cs_part, st)
                      : wgrad_launch<false, FT>(A, lda, B, ldb, M, N, K, sk, ks, kb, part, cs_part, st);
    if (rc) return rc;
    const long long total = (long long)M * (N >> 2) + (colsum_B ? (N >> 2) : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, part, nparts, M, N, C,
                       ldc, cs_part, colsum_B);
    return check_launch();
}

extern "C" int hsp_wgrad_f32(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
                             float* colsum_B, void* ws, size_t ws_bytes, hspStream_t stream) {
    return wgrad_impl<float>(A, lda, B, ldb, M, N, K, C, ldc, colsum_B, ws, ws_bytes, stream);
}
/* bf16 point rows in, fp32 parameter gradient out */
extern "C" int hsp_wgrad_bf16(const hsp_bf16_t* A, int lda, const hsp_bf16_t* B, int ldb, int M, int N, int K, float* C,
                              int ldc, float* colsum_B, void* ws, size_t ws_bytes, hspStream_t stream) {
    return wgrad_impl<bf16_t>(A, lda, B, ldb, M, N, K, C, ldc, colsum_B, ws, ws_bytes, stream);
}
