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

__global__ __launch_bounds__(256) void chamfer_nn_kernel(const float* __restrict__ a, int n,
                                                         const float* __restrict__ c, int m,
                                                         float* __restrict__ dist, int32_t* __restrict__ idx) {
    __shared__ float4 pts[CH_CHUNK];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* ab = a + (size_t)b * n * 3;
    const float* cb = c + (size_t)b * m * 3;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (i < n) { x1 = ab[i * 3]; y1 = ab[i * 3 + 1]; z1 = ab[i * 3 + 2]; }
    float best = 0.f;
    int besti = 0;
    for (int c0 = 0; c0 < m; c0 += CH_CHUNK) {
        const int cn = min(CH_CHUNK, m - c0);
        __syncthreads();
        for (int j = threadIdx.x; j < cn; j += 256)
            pts[j] = make_float4(cb[(c0 + j) * 3], cb[(c0 + j) * 3 + 1], cb[(c0 + j) * 3 + 2], 0.f);
        __syncthreads();
        if (i < n) {
            for (int j = 0; j < cn; ++j) {
                const float4 p = pts[j];
                const float dx = sub_rn(p.x, x1), dy = sub_rn(p.y, y1), dz = sub_rn(p.z, z1);
                const float d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
                if ((c0 + j) == 0 || d < best) { best = d; besti = c0 + j; }
            }
        }
    }
    if (i < n) {
        dist[(size_t)b * n + i] = best;
        idx[(size_t)b * n + i] = besti;
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
// FPS: one 1024-thread workgroup per cloud; dist-to-set lives in a global workspace (L2 resident);
// each round = distance update + workgroup arg-max (max value, then lowest index).
// ------------------------------------------------------------------------------------------------
#define FPS_THREADS 1024

__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ xyz, int N, int n_samples,
                                                          int32_t* __restrict__ sel, float* __restrict__ dts) {
    __shared__ float sv[FPS_THREADS / HSP_WAVE];
    __shared__ int si[FPS_THREADS / HSP_WAVE];
    __shared__ int scur;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    float* dt = dts + (size_t)b * N;
    for (int j = tid; j < N; j += FPS_THREADS) dt[j] = INFINITY;
    int cur = 0;
    for (int s = 0; s < n_samples; ++s) {
        if (tid == 0) sel[(size_t)b * n_samples + s] = cur;
        const float cx = p[cur * 3], cy = p[cur * 3 + 1], cz = p[cur * 3 + 2];
        float bv = -1.0f;
        int bi = INT_MAX;
        for (int j = tid; j < N; j += FPS_THREADS) {
            const float dx = sub_rn(p[j * 3], cx), dy = sub_rn(p[j * 3 + 1], cy), dz = sub_rn(p[j * 3 + 2], cz);
            const float d = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
            float v = dt[j];
            if (d < v) { v = d; dt[j] = d; }
            if (v > bv) { bv = v; bi = j; }          // j ascends per lane: strict '>' keeps the first max
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            float fv = sv[0];
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
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3((n + 255) / 256, B), dim3(256), 0, st, xyz1, n, xyz2, m, dist1, idx1);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3((m + 255) / 256, B), dim3(256), 0, st, xyz2, m, xyz1, n, dist2, idx2);
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
    hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(FPS_THREADS), 0, as_stream(stream), xyz, N, n_samples, sel,
                       reinterpret_cast<float*>(ws));
    return check_launch();
}
