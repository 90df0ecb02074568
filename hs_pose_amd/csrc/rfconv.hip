// rfconv.hip -- fused receptive-field graph convolution (3D-GCN style) for gfx950.
//
// Replaces HSlayer_surface.graph_conv / HS_layer.graph_conv of the reference
// (network/fs_net_repo/gcn3d.py:92-107, :158-181) together with the neighbour gather
// (gcn3d.py:39-47) and direction normalisation (gcn3d.py:49-59).  The reference materialises three
// (B,N,k,S*C) tensors per layer (1.18 GB each at B=16,N=1028,C=128); here a workgroup owns one
// point at a time, streams its k neighbour rows of the support tensor with 16-byte loads (the only
// large traffic, served from the XCD's L2: all workgroups of an XCD work on the same cloud), keeps
// the running max/arg-max in registers and writes only out (N,C) + arg-max bytes (N,S*C).
//
// Layout: column j of the S*C axis <-> (s = j / C, c = j % C)  (gcn3d.py:104,177).
// HBM-bound kernels: algorithmic bytes/point (fwd) = k*S*C*4 (gather, L2) + (S+1)*C*4/own row ... see DESIGN.md.
#include "common.h"

namespace hsp {

#define RF_THREADS 256

// point schedule shared by all rf kernels: blocks of XCD x handle clouds x, x+8, ... one cloud at a
// time (keeps that cloud's fm rows resident in the XCD-private 4 MiB L2); purely a speed choice.
struct PointIter {
    int b0, bstep, i0, istep;
    __device__ __forceinline__ PointIter(int B) {
        const int xcd = blockIdx.x % HSP_NUM_XCD;
        const int local = blockIdx.x / HSP_NUM_XCD;
        const int per_xcd = gridDim.x / HSP_NUM_XCD;
        if (B >= HSP_NUM_XCD) {            // several clouds per XCD, visited one after the other
            b0 = xcd; bstep = HSP_NUM_XCD; i0 = local; istep = per_xcd;
        } else {                           // fewer clouds than XCDs: XCDs x, x+B, ... share cloud x % B
            b0 = xcd % B; bstep = B * HSP_NUM_XCD;   // (visited once)
            const int share = xcd / B;
            const int nshare = (HSP_NUM_XCD - b0 + B - 1) / B;
            i0 = local + per_xcd * share; istep = per_xcd * nshare;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// forward.  SURFACE=true: out = mean_s max_n relu(z);  false: out = fm_c + mean_s max_n relu(z)*fm_support
// dynamic LDS: (S*C + 4*k + k) floats
// ------------------------------------------------------------------------------------------------
template <bool SURFACE>
__global__ __launch_bounds__(RF_THREADS) void rf_fwd_kernel(const float* __restrict__ xyz,
                                                            const int32_t* __restrict__ idx,
                                                            const float* __restrict__ dirs,
                                                            const float* __restrict__ fm, int B, int N, int k,
                                                            int S, int C, float* __restrict__ out,
                                                            uint8_t* __restrict__ argmax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int SC = S * C;
    float* smax = reinterpret_cast<float*>(smem);             // SC
    float4* sR = reinterpret_cast<float4*>(smax + SC);        // k  (unit direction, w unused)
    int* sIdx = reinterpret_cast<int*>(sR + k);               // k
    const int tid = threadIdx.x;
    const int nq = SC >> 2;                                   // float4 columns
    const int fstride = (S + 1) * C;
    const float invS_div = (float)S;
    const PointIter it(B);
    for (int b = it.b0; b < B; b += it.bstep) {
        const float* xb = xyz + (size_t)b * N * 3;
        for (int i = it.i0; i < N; i += it.istep) {
            const size_t pt = (size_t)b * N + i;
            __syncthreads();                                   // previous point's LDS reads are done
            if (tid < k) {
                const int m = idx[pt * k + tid];
                sIdx[tid] = m;
                const float3 r = unit_dir(xb[i * 3], xb[i * 3 + 1], xb[i * 3 + 2], xb[m * 3], xb[m * 3 + 1], xb[m * 3 + 2]);
                sR[tid] = make_float4(r.x, r.y, r.z, 0.f);
            }
            __syncthreads();
            for (int cq = tid; cq < nq; cq += RF_THREADS) {
                const int j = cq << 2;
                const float4 d0 = *reinterpret_cast<const float4*>(dirs + j);
                const float4 d1 = *reinterpret_cast<const float4*>(dirs + SC + j);
                const float4 d2 = *reinterpret_cast<const float4*>(dirs + 2 * SC + j);
                float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                const float* fsup = SURFACE ? nullptr : fm + (size_t)b * N * fstride + C + j;
#pragma unroll 4
                for (int n = 0; n < k; ++n) {
                    const float4 r = sR[n];
                    // theta = relu(R . D) with the k-ordered fma chain of the reference's matmul
                    float4 th;
                    th.x = fmaxf(__fmaf_rn(r.z, d2.x, __fmaf_rn(r.y, d1.x, mul_rn(r.x, d0.x))), 0.f);
                    th.y = fmaxf(__fmaf_rn(r.z, d2.y, __fmaf_rn(r.y, d1.y, mul_rn(r.x, d0.y))), 0.f);
                    th.z = fmaxf(__fmaf_rn(r.z, d2.z, __fmaf_rn(r.y, d1.z, mul_rn(r.x, d0.z))), 0.f);
                    th.w = fmaxf(__fmaf_rn(r.z, d2.w, __fmaf_rn(r.y, d1.w, mul_rn(r.x, d0.w))), 0.f);
                    if (!SURFACE) {
                        const float4 f = *reinterpret_cast<const float4*>(fsup + (size_t)sIdx[n] * fstride);
                        th.x = mul_rn(th.x, f.x); th.y = mul_rn(th.y, f.y);
                        th.z = mul_rn(th.z, f.z); th.w = mul_rn(th.w, f.w);
                    }
                    if (th.x > best.x) { best.x = th.x; a0 = n; }
                    if (th.y > best.y) { best.y = th.y; a1 = n; }
                    if (th.z > best.z) { best.z = th.z; a2 = n; }
                    if (th.w > best.w) { best.w = th.w; a3 = n; }
                }
                *reinterpret_cast<float4*>(smax + j) = best;
                *reinterpret_cast<uchar4*>(argmax + pt * SC + j) = make_uchar4((unsigned char)a0, (unsigned char)a1, (unsigned char)a2, (unsigned char)a3);
            }
            __syncthreads();
            for (int c = tid; c < C; c += RF_THREADS) {
                float s = smax[c];
                for (int sp = 1; sp < S; ++sp) s = add_rn(s, smax[sp * C + c]);
                float v = __fdiv_rn(s, invS_div);
                if (!SURFACE) v = add_rn(fm[pt * fstride + c], v);
                out[pt * C + c] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward (scatter form).  For every point i and column j: n = argmax, m = idx[i][n];
//   grad_fm[b,m,C+j] += ga * theta        (atomic; support rows are shared between points)
//   grad_fm[b,i,c]    = g[b,i,c]          (centre, plain store; the buffer was zeroed before)
//   gD[d][j]         += ga * fm[b,m,C+j] * [z>0] * R[n][d]   (registers -> per-block partials in ws)
// with ga = g[b,i,j%C] / S.  SURFACE: only gD, with fm := 1.
// ws layout: [gridDim.x][3][SC] floats of partial direction gradients, reduced by rf_dirs_reduce_kernel.
// dynamic LDS: (C + 4*k + k) floats
// ------------------------------------------------------------------------------------------------
template <bool SURFACE, int NCH>
__global__ __launch_bounds__(RF_THREADS) void rf_bwd_kernel(const float* __restrict__ xyz,
                                                            const int32_t* __restrict__ idx,
                                                            const float* __restrict__ dirs,
                                                            const float* __restrict__ fm,
                                                            const uint8_t* __restrict__ argmax,
                                                            const float* __restrict__ gout, int B, int N, int k,
                                                            int S, int C, float* __restrict__ gfm,
                                                            float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int SC = S * C;
    float* sg = reinterpret_cast<float*>(smem);               // C : g / S
    float4* sR = reinterpret_cast<float4*>(sg + C);           // k
    int* sIdx = reinterpret_cast<int*>(sR + k);               // k
    const int tid = threadIdx.x;
    const int nq = SC >> 2;
    const int fstride = (S + 1) * C;
    const float Sdiv = (float)S;
    // per-thread direction-gradient accumulators for its (up to NCH) float4 column groups
    float4 gd0[NCH], gd1[NCH], gd2[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) gd0[u] = gd1[u] = gd2[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const PointIter it(B);
    for (int b = it.b0; b < B; b += it.bstep) {
        const float* xb = xyz + (size_t)b * N * 3;
        for (int i = it.i0; i < N; i += it.istep) {
            const size_t pt = (size_t)b * N + i;
            __syncthreads();
            if (tid < k) {
                const int m = idx[pt * k + tid];
                sIdx[tid] = m;
                const float3 r = unit_dir(xb[i * 3], xb[i * 3 + 1], xb[i * 3 + 2], xb[m * 3], xb[m * 3 + 1], xb[m * 3 + 2]);
                sR[tid] = make_float4(r.x, r.y, r.z, 0.f);
            }
            for (int c = tid; c < C; c += RF_THREADS) {
                const float g = gout[pt * C + c];
                sg[c] = __fdiv_rn(g, Sdiv);
                if (!SURFACE) gfm[pt * fstride + c] = g;   // centre columns: dense, owned by this point
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                const int cq = tid + u * RF_THREADS;
                if (cq < nq) {
                    const int j = cq << 2;
                    const uchar4 am = *reinterpret_cast<const uchar4*>(argmax + pt * SC + j);
                    const int c = j % C;                     // C % 4 == 0: the 4 columns share s
                    const float4 d0 = *reinterpret_cast<const float4*>(dirs + j);
                    const float4 d1 = *reinterpret_cast<const float4*>(dirs + SC + j);
                    const float4 d2 = *reinterpret_cast<const float4*>(dirs + 2 * SC + j);
                    const float4 ga = *reinterpret_cast<const float4*>(sg + c);
                    const unsigned char an[4] = {am.x, am.y, am.z, am.w};
                    const float gav[4] = {ga.x, ga.y, ga.z, ga.w};
                    const float d0v[4] = {d0.x, d0.y, d0.z, d0.w};
                    const float d1v[4] = {d1.x, d1.y, d1.z, d1.w};
                    const float d2v[4] = {d2.x, d2.y, d2.z, d2.w};
                    float* a0p = reinterpret_cast<float*>(&gd0[u]);
                    float* a1p = reinterpret_cast<float*>(&gd1[u]);
                    float* a2p = reinterpret_cast<float*>(&gd2[u]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = an[e];
                        const float4 r = sR[n];
                        const float z = __fmaf_rn(r.z, d2v[e], __fmaf_rn(r.y, d1v[e], mul_rn(r.x, d0v[e])));
                        float w = gav[e];
                        if (!SURFACE) {
                            const size_t off = ((size_t)b * N + sIdx[n]) * fstride + C + j + e;
                            const float fv = fm[off];
                            const float th = fmaxf(z, 0.f);
                            if (th != 0.f && gav[e] != 0.f) atomicAdd(gfm + off, gav[e] * th);
                            w = gav[e] * fv;
                        }
                        if (z > 0.f) {
                            a0p[e] += w * r.x;
                            a1p[e] += w * r.y;
                            a2p[e] += w * r.z;
                        }
                    }
                }
            }
        }
    }
    // per-block partials
    float* wsb = ws + (size_t)blockIdx.x * 3 * SC;
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int cq = tid + u * RF_THREADS;
        if (cq < nq) {
            const int j = cq << 2;
            *reinterpret_cast<float4*>(wsb + j) = gd0[u];
            *reinterpret_cast<float4*>(wsb + SC + j) = gd1[u];
            *reinterpret_cast<float4*>(wsb + 2 * SC + j) = gd2[u];
        }
    }
}

// out[e] = sum over blocks of ws[blk][e], e in [0, 3*SC): fixed order => deterministic
__global__ __launch_bounds__(256) void rf_dirs_reduce_kernel(const float* __restrict__ ws, int nblk, int n3,
                                                             float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n3) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += ws[(size_t)b * n3 + e];
    out[e] = s;
}

// backward grid: persistent, capped so that the per-block direction-gradient partials stay <= 8 MiB
static int rf_bwd_grid_max(int SC) {
    long long g = (8ll << 20) / (12ll * SC);
    g &= ~7ll;
    if (g < 8) g = 8;
    if (g > 1024) g = 1024;
    return (int)g;
}

static int rf_bwd_grid(long long points, int SC) {
    const int g = persistent_blocks(points, 4), gm = rf_bwd_grid_max(SC);
    return g < gm ? g : gm;
}

}  // namespace hsp

using namespace hsp;

static int rf_check(const void* a, const void* b, const void* c, int B, int N, int k, int S, int C) {
    if (!a || !b || !c || B <= 0 || N <= 0 || k <= 0 || S <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    if (k > 255 || (C & 3)) return HSP_ERR_UNSUPPORTED;   // arg-max is a byte; float4 column groups
    return HSP_OK;
}

template <bool SURFACE>
static int rf_fwd(const float* xyz, const int32_t* idx, const float* dirs, const float* fm, int B, int N, int k,
                  int S, int C, float* out, uint8_t* argmax, hspStream_t stream) {
    int rc = rf_check(xyz, idx, dirs, B, N, k, S, C);
    if (rc) return rc;
    if (!out || !argmax || (!SURFACE && !fm)) return HSP_ERR_BAD_ARG;
    const size_t lds = (size_t)(S * C + 5 * k) * 4;
    if (lds > 64 * 1024) return HSP_ERR_UNSUPPORTED;
    const int grid = persistent_blocks((long long)B * N, 8);
    hipLaunchKernelGGL(rf_fwd_kernel<SURFACE>, dim3(grid), dim3(RF_THREADS), lds, as_stream(stream), xyz, idx, dirs,
                       fm, B, N, k, S, C, out, argmax);
    return check_launch();
}

extern "C" int hsp_rf_surface_fwd(const float* xyz, const int32_t* idx, const float* dirs_n, int B, int N, int k,
                                  int S, int K, float* out, uint8_t* argmax, hspStream_t stream) {
    return rf_fwd<true>(xyz, idx, dirs_n, nullptr, B, N, k, S, K, out, argmax, stream);
}

extern "C" int hsp_rf_conv_fwd(const float* xyz, const int32_t* idx, const float* dirs_n, const float* fm, int B,
                               int N, int k, int S, int C, float* out, uint8_t* argmax, hspStream_t stream) {
    return rf_fwd<false>(xyz, idx, dirs_n, fm, B, N, k, S, C, out, argmax, stream);
}

extern "C" size_t hsp_rf_bwd_workspace_bytes(int SC) {
    if (SC <= 0) return 0;
    return (size_t)rf_bwd_grid_max(SC) * 3 * (size_t)SC * sizeof(float);
}

template <bool SURFACE>
static int rf_bwd(const float* xyz, const int32_t* idx, const float* dirs, const float* fm, const uint8_t* argmax,
                  const float* gout, int B, int N, int k, int S, int C, float* gfm, float* gdirs, void* ws,
                  size_t ws_bytes, hspStream_t stream) {
    int rc = rf_check(xyz, idx, dirs, B, N, k, S, C);
    if (rc) return rc;
    if (!argmax || !gout || !gdirs || (!SURFACE && (!fm || !gfm))) return HSP_ERR_BAD_ARG;
    const int SC = S * C;
    if (!ws || ws_bytes < hsp_rf_bwd_workspace_bytes(SC)) return HSP_ERR_WORKSPACE;
    const int nq = SC >> 2;
    const int nch = (nq + RF_THREADS - 1) / RF_THREADS;
    if (nch > 4) return HSP_ERR_UNSUPPORTED;              // S*C <= 4096
    hipStream_t st = as_stream(stream);
    if (!SURFACE) {
        hipError_t e = hipMemsetAsync(gfm, 0, (size_t)B * N * (S + 1) * C * sizeof(float), st);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    const int grid = rf_bwd_grid((long long)B * N, SC);
    const size_t lds = (size_t)(C + 5 * k) * 4;
    float* wsf = reinterpret_cast<float*>(ws);
#define RF_BWD_LAUNCH(NCH)                                                                                     \
    hipLaunchKernelGGL((rf_bwd_kernel<SURFACE, NCH>), dim3(grid), dim3(RF_THREADS), lds, st, xyz, idx, dirs, fm, \
                       argmax, gout, B, N, k, S, C, gfm, wsf)
    switch (nch) {
        case 1: RF_BWD_LAUNCH(1); break;
        case 2: RF_BWD_LAUNCH(2); break;
        case 3: RF_BWD_LAUNCH(3); break;
        default: RF_BWD_LAUNCH(4); break;
    }
#undef RF_BWD_LAUNCH
    rc = check_launch();
    if (rc) return rc;
    const int n3 = 3 * SC;
    hipLaunchKernelGGL(rf_dirs_reduce_kernel, dim3((n3 + 255) / 256), dim3(256), 0, st, wsf, grid, n3, gdirs);
    return check_launch();
}

extern "C" int hsp_rf_surface_bwd(const float* xyz, const int32_t* idx, const float* dirs_n, const uint8_t* argmax,
                                  const float* grad_out, int B, int N, int k, int S, int K, float* grad_dirs_n,
                                  void* ws, size_t ws_bytes, hspStream_t stream) {
    return rf_bwd<true>(xyz, idx, dirs_n, nullptr, argmax, grad_out, B, N, k, S, K, nullptr, grad_dirs_n, ws,
                        ws_bytes, stream);
}

extern "C" int hsp_rf_conv_bwd(const float* xyz, const int32_t* idx, const float* dirs_n, const float* fm,
                               const uint8_t* argmax, const float* grad_out, int B, int N, int k, int S, int C,
                               float* grad_fm, float* grad_dirs_n, void* ws, size_t ws_bytes, hspStream_t stream) {
    return rf_bwd<false>(xyz, idx, dirs_n, fm, argmax, grad_out, B, N, k, S, C, grad_fm, grad_dirs_n, ws, ws_bytes,
                         stream);
}
