// chamfer_fps.hip -- Chamfer distance (fwd/bwd) and farthest point sampling for gfx950.
//
// Chamfer: replaces the reference's CUDA extension tools/pyTorchChamferDistance/chamfer_distance.cu
// (:6-137 forward, :158-187 backward) / its CPU twin chamfer_distance.cpp:59-177.  Same results:
// squared distance from coordinate differences in fp32, FIRST minimum wins (strict '<').
// Design differs from the reference kernel: one lane per query point, the other cloud staged in
// LDS as float4 in chunks of 2048 points (broadcast ds_read_b128, no bank conflicts), grid sized
// from the problem instead of a fixed dim3(32,16).
// FPS: replaces tools/eval_utils.py:107-119 (per cloud): start at 0, running min, first max wins.
#include "common.h"

namespace hsp {

#define CH_CHUNK 2048

// 64 queries per workgroup (lane = query); the 4 waves scan the 4 quarters of every staged chunk and the partial
// (distance, index) pairs meet in LDS -- lexicographic minimum == the first minimum of a serial scan.  With one wave per
// 64 queries a 1028-point cloud pair gives 272 waves for 1024 SIMDs; this form gives 1088.
__global__ __launch_bounds__(256) void chamfer_nn_kernel(const float* __restrict__ a, int n,
                                                         const float* __restrict__ c, int m,
                                                         float* __restrict__ dist, int32_t* __restrict__ idx) {
    __shared__ float4 pts[CH_CHUNK];
    __shared__ float sd[4][64];
    __shared__ int si[4][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * 64 + lane;
    const float* ab = a + (size_t)b * n * 3;
    const float* cb = c + (size_t)b * m * 3;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (i < n) { x1 = ab[i * 3]; y1 = ab[i * 3 + 1]; z1 = ab[i * 3 + 2]; }
    float best = INFINITY;
    int besti = INT_MAX;
    for (int c0 = 0; c0 < m; c0 += CH_CHUNK) {
        const int cn = min(CH_CHUNK, m - c0);
        __syncthreads();
        for (int j = threadIdx.x; j < cn; j += 256)
            pts[j] = make_float4(cb[(c0 + j) * 3], cb[(c0 + j) * 3 + 1], cb[(c0 + j) * 3 + 2], 0.f);
        __syncthreads();
        const int q = (cn + 3) >> 2;
        const int j0 = wv * q, j1 = min(cn, j0 + q);
        int j = j0;
        for (; j + 7 < j1; j += 8) {                         // 8 broadcast LDS reads in flight per round trip
            float4 p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = pts[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float dx = sub_rn(p[u].x, x1), dy = sub_rn(p[u].y, y1), dz = sub_rn(p[u].z, z1);
                const float d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
                if (d < best) { best = d; besti = c0 + j + u; }
            }
        }
        for (; j < j1; ++j) {
            const float4 p = pts[j];
            const float dx = sub_rn(p.x, x1), dy = sub_rn(p.y, y1), dz = sub_rn(p.z, z1);
            const float d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
            if (d < best) { best = d; besti = c0 + j; }
        }
    }
    sd[wv][lane] = best;
    si[wv][lane] = besti;
    __syncthreads();
    if (wv == 0 && i < n) {
        // lexicographic minimum of the partial (distance, index) pairs == the serial scan's first minimum
        float bd = sd[0][lane];
        int bi = si[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float d = sd[w][lane];
            const int k = si[w][lane];
            if (d < bd || (d == bd && k < bi)) { bd = d; bi = k; }
        }
        if (bi == INT_MAX) {        // nothing compared below +inf (NaN / overflowing input): the reference keeps candidate 0
            const float dx = sub_rn(cb[0], x1), dy = sub_rn(cb[1], y1), dz = sub_rn(cb[2], z1);
            bd = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
            bi = 0;
        }
        dist[(size_t)b * n + i] = bd;
        idx[(size_t)b * n + i] = bi;
    }
}

// one direction of the backward: ga += g*(p-q), gc[idx] -= g*(p-q), g = 2*gd   (both pre-zeroed)
__global__ __launch_bounds__(256) void chamfer_grad_kernel(const float* __restrict__ a, int n,
                                                           const float* __restrict__ c, int m,
                                                           const int32_t* __restrict__ idx,
                                                           const float* __restrict__ gd, float* __restrict__ ga,
                                                           float* __restrict__ gc) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t pi = (size_t)b * n + i;
    const int j2 = idx[pi];
    const size_t pj = (size_t)b * m + j2;
    const float g = mul_rn(gd[pi], 2.0f);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float v = mul_rn(g, sub_rn(a[pi * 3 + d], c[pj * 3 + d]));
        atomicAdd(ga + pi * 3 + d, v);
        atomicAdd(gc + pj * 3 + d, -v);
    }
}

// ------------------------------------------------------------------------------------------------
// FPS.  A pick is a dependent chain (distance update -> arg-max over the cloud -> next centre), so the kernel is
// built around the length of that chain: one workgroup per cloud, the cloud's coordinates AND the running
// distance-to-set in registers (PPT points per thread, point j = tid + k * THREADS so indices ascend per lane),
// a copy of the coordinates in LDS for the centre look-up, and per pick
//   registers: update + per-lane best  ->  one 64-bit key (value bits | ~index: max = largest value, lowest index)
//   wave:      4 DPP steps inside the 16-lane rows + 4 v_readlane across the rows
//   workgroup: one LDS slot per wave, ONE barrier (slots double-buffered by pick parity), every thread folds
//              the <= 16 slots itself and reads the winner's coordinates from LDS (broadcast).
// No global memory inside the loop except the 4-byte result store.  (fps_generic_kernel below: any N, distances
// in a global workspace.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long k, const int ctrl_sel) {
    unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32), olo, ohi;
    switch (ctrl_sel) {                                     // dpp_ctrl must be an immediate
        case 0: olo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
                ohi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;
        case 1: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
                ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;
        case 2: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false);     // row_half_mirror
                ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break;
        default: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false);    // row_mirror
                 ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break;
    }
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o > k ? o : k;
}

template <int THREADS, int PPT>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(const float* __restrict__ xyz, int N, int n_samples,
                                                          int32_t* __restrict__ sel) {
    constexpr int W = THREADS / HSP_WAVE;
    extern __shared__ __attribute__((aligned(16))) float s_xyz[];          // N * 3
    __shared__ unsigned long long slot[2][W];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* p = xyz + (size_t)b * N * 3;
    for (int e = tid; e < N * 3; e += THREADS) s_xyz[e] = p[e];
    float px[PPT], py[PPT], pz[PPT], dt[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int j = tid + k * THREADS;
        const bool in = j < N;
        px[k] = in ? p[j * 3] : 0.f;
        py[k] = in ? p[j * 3 + 1] : 0.f;
        pz[k] = in ? p[j * 3 + 2] : 0.f;
        dt[k] = INFINITY;
    }
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < n_samples; ++s) {
        if (tid == 0) sel[(size_t)b * n_samples + s] = cur;
        const float cx = s_xyz[cur * 3], cy = s_xyz[cur * 3 + 1], cz = s_xyz[cur * 3 + 2];
        unsigned long long key = 0;                         // (value bits << 32) | ~index; values are >= 0
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int j = tid + k * THREADS;
            const float dx = sub_rn(px[k], cx), dy = sub_rn(py[k], cy), dz = sub_rn(pz[k], cz);
            // numpy on a float32 cloud: fp32 (x*x + y*y) + z*z, then a correctly rounded fp32 sqrt (eval_utils.py:73-84);
            // the sqrt folds neighbouring d^2 values into exact ties that argmax breaks by index, so it cannot be skipped
            const float d = sqrtf(add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz)));   // (__fsqrt_rn is the 1-ulp native sqrt)
            const float v = fminf(dt[k], d);
            dt[k] = v;
            const unsigned long long kk = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(~j);
            if (j < N && kk > key) key = kk;
        }
        key = dpp_max_u64(key, 0);
        key = dpp_max_u64(key, 1);
        key = dpp_max_u64(key, 2);
        key = dpp_max_u64(key, 3);                          // every lane holds its 16-lane row's best
        unsigned long long wk = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)key, r * 16);
            const unsigned hi = __builtin_amdgcn_readlane((unsigned)(key >> 32), r * 16);
            const unsigned long long rk = ((unsigned long long)hi << 32) | lo;
            wk = rk > wk ? rk : wk;
        }
        if (W > 1) {
            if (lane == 0) slot[s & 1][wv] = wk;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const unsigned long long o = slot[s & 1][w];
                wk = o > wk ? o : wk;
            }
        }
        cur = (int)(~(unsigned)wk);
    }
}

// any N: one 1024-thread workgroup per cloud; dist-to-set lives in a global workspace (L2 resident);
// each round = distance update + workgroup arg-max (max value, then lowest index).
#define FPS_THREADS 1024

template <typename T> __device__ __forceinline__ T fps_dist(T dx, T dy, T dz);
template <> __device__ __forceinline__ float fps_dist<float>(float dx, float dy, float dz) {
    return sqrtf(add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz)));      // correctly rounded (ocml)
}
template <> __device__ __forceinline__ double fps_dist<double>(double dx, double dy, double dz) {
    return sqrt(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
}

// T = float: any N (distances in a global workspace); T = double: the reference helper's own dtype (it is called on
// float64 mesh vertices, tools/eval_utils.py:122-140) -- numpy's float64 arithmetic step for step.
template <typename T>
__global__ __launch_bounds__(FPS_THREADS) void fps_generic_kernel(const T* __restrict__ xyz, int N, int n_samples,
                                                                  int32_t* __restrict__ sel, T* __restrict__ dts) {
    __shared__ T sv[FPS_THREADS / HSP_WAVE];
    __shared__ int si[FPS_THREADS / HSP_WAVE];
    __shared__ int scur;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const T* p = xyz + (size_t)b * N * 3;
    T* dt = dts + (size_t)b * N;
    for (int j = tid; j < N; j += FPS_THREADS) dt[j] = (T)INFINITY;
    int cur = 0;
    for (int s = 0; s < n_samples; ++s) {
        if (tid == 0) sel[(size_t)b * n_samples + s] = cur;
        const T cx = p[cur * 3], cy = p[cur * 3 + 1], cz = p[cur * 3 + 2];
        T bv = (T)-1.0;
        int bi = INT_MAX;
        for (int j = tid; j < N; j += FPS_THREADS) {
            const T d = fps_dist<T>(p[j * 3] - cx, p[j * 3 + 1] - cy, p[j * 3 + 2] - cz);
            T v = dt[j];
            if (d < v) { v = d; dt[j] = d; }
            if (v > bv) { bv = v; bi = j; }          // j ascends per lane: strict '>' keeps the first max
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const T ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            T fv = sv[0];
            int fi = si[0];
            for (int w = 1; w < FPS_THREADS / HSP_WAVE; ++w)
                if (sv[w] > fv || (sv[w] == fv && si[w] < fi)) { fv = sv[w]; fi = si[w]; }
            scur = fi;
        }
        __syncthreads();
        cur = scur;
    }
}

}  // namespace hsp

using namespace hsp;

extern "C" int hsp_chamfer_fwd(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1, float* dist2,
                               int32_t* idx1, int32_t* idx2, hspStream_t stream) {
    if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || B <= 0 || n <= 0 || m <= 0) return HSP_ERR_BAD_ARG;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3((n + 63) / 64, B), dim3(256), 0, st, xyz1, n, xyz2, m, dist1, idx1);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3((m + 63) / 64, B), dim3(256), 0, st, xyz2, m, xyz1, n, dist2, idx2);
    return check_launch();
}

extern "C" int hsp_chamfer_bwd(const float* xyz1, const float* xyz2, const int32_t* idx1, const int32_t* idx2,
                               const float* gd1, const float* gd2, int B, int n, int m, float* gx1, float* gx2,
                               hspStream_t stream) {
    if (!xyz1 || !xyz2 || !idx1 || !idx2 || !gd1 || !gd2 || !gx1 || !gx2 || B <= 0 || n <= 0 || m <= 0)
        return HSP_ERR_BAD_ARG;
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemsetAsync(gx1, 0, (size_t)B * n * 3 * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(gx2, 0, (size_t)B * m * 3 * sizeof(float), st);
    if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3((n + 255) / 256, B), dim3(256), 0, st, xyz1, n, xyz2, m, idx1, gd1, gx1, gx2);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3((m + 255) / 256, B), dim3(256), 0, st, xyz2, m, xyz1, n, idx2, gd2, gx2, gx1);
    return check_launch();
}

extern "C" size_t hsp_fps_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * N * sizeof(float);
}

extern "C" int hsp_fps_f32(const float* xyz, int B, int N, int n_samples, int32_t* sel, void* ws, size_t ws_bytes,
                           hspStream_t stream) {
    if (!xyz || !sel || B <= 0 || N <= 0 || n_samples <= 0 || n_samples > N) return HSP_ERR_BAD_ARG;
    if (!ws || ws_bytes < hsp_fps_workspace_bytes(B, N)) return HSP_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    const size_t lds = (size_t)N * 3 * sizeof(float);
#define FPS_REG(THREADS, PPT)                                                                                          \
    do {                                                                                                               \
        auto kern = fps_reg_kernel<THREADS, PPT>;                                                                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }                                         \
        hipLaunchKernelGGL(kern, dim3(B), dim3(THREADS), lds, st, xyz, N, n_samples, sel);                             \
        return check_launch();                                                                                         \
    } while (0)
    if (N <= 12288) {                                          // coordinates fit LDS (144 KB) and a few dozen registers
        if (N <= 64 * 2) FPS_REG(64, 2);
        if (N <= 256 * 2) FPS_REG(256, 2);
        if (N <= 256 * 5) FPS_REG(256, 5);                     // N = 1028 (128 x 10 and 512 x 3: 15 % slower)
        if (N <= 256 * 8) FPS_REG(256, 8);
        if (N <= 256 * 16) FPS_REG(256, 16);                   // N = 4096 (512 x 8 is as fast, 1024 x 4 is 40 % slower)
        if (N <= 1024 * 8) FPS_REG(1024, 8);
        FPS_REG(1024, 12);
    }
#undef FPS_REG
    hipLaunchKernelGGL(fps_generic_kernel<float>, dim3(B), dim3(FPS_THREADS), 0, st, xyz, N, n_samples, sel,
                       reinterpret_cast<float*>(ws));
    return check_launch();
}

extern "C" int hsp_fps_f64(const double* xyz, int B, int N, int n_samples, int32_t* sel, void* ws, size_t ws_bytes,
                           hspStream_t stream) {
    if (!xyz || !sel || B <= 0 || N <= 0 || n_samples <= 0 || n_samples > N) return HSP_ERR_BAD_ARG;
    if (!ws || ws_bytes < 2 * hsp_fps_workspace_bytes(B, N)) return HSP_ERR_WORKSPACE;
    hipLaunchKernelGGL(fps_generic_kernel<double>, dim3(B), dim3(FPS_THREADS), 0, as_stream(stream), xyz, N, n_samples,
                       sel, reinterpret_cast<double*>(ws));
    return check_launch();
}
