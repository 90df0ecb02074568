// norm.hip -- fused train-mode BatchNorm1d + ReLU over point rows for gfx950.
//
// Replaces  F.relu(bn(x.transpose(1,2)).transpose(1,2))  of the HS stack (reference
// network/fs_net_repo/FaceRecon.py:27-29, :90-95): BatchNorm1d over the channel dim of a (B,N,C)
// tensor with batch statistics over the R = B*N rows, eps 1e-5, momentum 0.1, followed by ReLU.
// The reference transposes to (B,C,N) and back; here rows stay point-major and one lane owns 4
// channels, so every access is a 16-byte row segment.
//
//   forward   partial shifted sums per row chunk -> finalize (mean, invstd, running-stat update,
//             num_batches_tracked) -> apply + ReLU                        (3 launches, x read twice)
//   backward  dz = dy*[y>0]; partial (sum dz, sum dz*xhat) -> finalize (d gamma, d beta)
//             -> dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat))   (3 launches)
// Sums are "shifted" by the first row of the tensor (sum (x-s), sum (x-s)^2), which keeps fp32
// cancellation harmless when |mean| >> std; partials are folded in a fixed order (deterministic).
// HBM-bound: forward algorithmic bytes = 4*R*C in + 4*R*C out; backward 8*R*C in + 4*R*C out.
#include "common.h"

namespace hsp {

#define BN_THREADS 256

// four consecutive values of a row whose pitch is only 8-byte aligned (an even fp32 pitch such as feat's 1286: a gradient
// column block of a dense (B, N, 1286) tensor): two 8-byte loads
template <typename FT>
__device__ __forceinline__ float4 bn_ld4_pitch(const FT* p, int ld) {
    if constexpr (sizeof(FT) == 4) {
        if (ld & 3) {
            const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
            return make_float4(a.x, a.y, b.x, b.y);
        }
    }
    return Feat<FT>::ld4(p);
}
#define BN_MAX_PARTIALS 512       // row chunks (workgroups of the partial kernels); folded 16-way parallel by finalize

// partial[blk][0][c] = sum_r v1, partial[blk][1][c] = sum_r v2 over the rows of chunk blk, where
//   MODE 0 (forward stats):  v1 = x - shift,  v2 = (x - shift)^2            shift = x[0][c]
//   MODE 1 (backward):       v1 = dz,         v2 = dz * xhat                dz = relu ? dy*[a>0] : dy
// FT: storage type of the row tensors x / dy / y / dx (statistics, affine parameters and partial sums are fp32)
// XT: storage type of x (the BatchNorm INPUT) -- fp32 with bf16 y / dy / dx in the "mixed" form
template <int MODE, typename FT, typename XT>
__global__ __launch_bounds__(BN_THREADS) void bn_partial_kernel(const XT* __restrict__ x,
                                                                const FT* __restrict__ dy, int R, int C,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int relu,
                                                                float* __restrict__ partial, int rows_per_block,
                                                                int ldy = 0, const FT* __restrict__ dy2 = nullptr, int ldy2 = 0) {
    __shared__ float4 red[2][BN_THREADS];
    const int cq = C >> 2;                       // float4 groups per row
    const int tid = threadIdx.x;
    const int g = tid % cq, rl = tid / cq;       // BN_THREADS % cq == 0 is guaranteed by the launcher
    const int RL = BN_THREADS / cq;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(R, r0 + rows_per_block);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), mu = sh, is = sh, ga = sh, be = sh;
    if (MODE == 0) {
        sh = Feat<XT>::ld4(x + (g << 2));
    } else {
        mu = *reinterpret_cast<const float4*>(mean + (g << 2));
        is = *reinterpret_cast<const float4*>(invstd + (g << 2));
        ga = *reinterpret_cast<const float4*>(gamma + (g << 2));
        be = *reinterpret_cast<const float4*>(beta + (g << 2));
    }
    for (int r = r0 + rl; r < r1; r += RL) {
        const float4 v = Feat<XT>::ld4(x + (size_t)r * C + (g << 2));
        if (MODE == 0) {
            const float a = v.x - sh.x, b = v.y - sh.y, c = v.z - sh.z, d = v.w - sh.w;
            s1.x += a; s1.y += b; s1.z += c; s1.w += d;
            s2.x += a * a; s2.y += b * b; s2.z += c * c; s2.w += d * d;
        } else {
            // dy: rows of pitch ldy (a column block of a wider gradient tensor is consumed in place); dy2: a second incoming
            // gradient of the same tensor (two consumers: the sum autograd would form with a separate kernel)
            float4 dz = bn_ld4_pitch<FT>(dy + (size_t)r * (ldy ? ldy : C) + (g << 2), ldy);
            if (dy2) {
                const float4 d2 = bn_ld4_pitch<FT>(dy2 + (size_t)r * ldy2 + (g << 2), ldy2);
                dz.x += d2.x; dz.y += d2.y; dz.z += d2.z; dz.w += d2.w;
            }
            const float4 xh = make_float4((v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w);
            if (relu) {
                if (!(xh.x * ga.x + be.x > 0.f)) dz.x = 0.f;
                if (!(xh.y * ga.y + be.y > 0.f)) dz.y = 0.f;
                if (!(xh.z * ga.z + be.z > 0.f)) dz.z = 0.f;
                if (!(xh.w * ga.w + be.w > 0.f)) dz.w = 0.f;
            }
            s1.x += dz.x; s1.y += dz.y; s1.z += dz.z; s1.w += dz.w;
            s2.x += dz.x * xh.x; s2.y += dz.y * xh.y; s2.z += dz.z * xh.z; s2.w += dz.w * xh.w;
        }
    }
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < RL; ++l) {          // fixed order
            const float4 a = red[0][l * cq + g], b = red[1][l * cq + g];
            s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
            s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
        }
        float* p = partial + (size_t)blockIdx.x * 2 * C;
        *reinterpret_cast<float4*>(p + (g << 2)) = s1;
        *reinterpret_cast<float4*>(p + C + (g << 2)) = s2;
    }
}

// one thread per channel: fold the partials (ascending block order), then
//   MODE 0: mean, biased var -> invstd; running stats (momentum, unbiased var); num_batches_tracked += 1
//   MODE 1: dgamma = sum dz*xhat, dbeta = sum dz; also keep both means for the dx pass
template <int MODE, typename FT>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int R, int C,
                                                           const FT* __restrict__ x, float eps, float momentum,
                                                           float* __restrict__ out_a, float* __restrict__ out_b,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var,
                                                           long long* __restrict__ num_batches) {
    // workgroup = 16 channels x 64 slices of the partials: every thread sums its <= BN_MAX_PARTIALS/64 = 8 partials with all
    // loads in flight (one L2 round trip; a single thread walking 64 partials is a 10 us chain), then the 64 slices are
    // folded through LDS in a fixed order (deterministic).  (16 x 64 rather than 64 x 16: four times the workgroups --
    // C = 128 gave two -- and a quarter of the dependent rounds: 8.2 -> 4 us at 16448 rows.)
    __shared__ float red[64][2][16];
    const int lc = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + lc;
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
        float u1[4] = {0.f, 0.f, 0.f, 0.f}, u2[4] = {0.f, 0.f, 0.f, 0.f};
        int b = sl;
        for (; b + 192 < nblk; b += 256) {                  // 8 loads in flight
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                u1[u] += partial[(size_t)(b + 64 * u) * 2 * C + c];
                u2[u] += partial[(size_t)(b + 64 * u) * 2 * C + C + c];
            }
        }
        for (; b < nblk; b += 64) {
            u1[0] += partial[(size_t)b * 2 * C + c];
            u2[0] += partial[(size_t)b * 2 * C + C + c];
        }
        a1 = (u1[0] + u1[1]) + (u1[2] + u1[3]);
        a2 = (u2[0] + u2[1]) + (u2[2] + u2[3]);
    }
    red[sl][0][lc] = a1;
    red[sl][1][lc] = a2;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    float s1 = red[0][0][lc], s2 = red[0][1][lc];
#pragma unroll 8
    for (int t = 1; t < 64; ++t) { s1 += red[t][0][lc]; s2 += red[t][1][lc]; }
    if (MODE == 0) {
        const float invR = 1.0f / (float)R;
        const float ms = s1 * invR;                              // mean of (x - shift)
        float var = s2 * invR - ms * ms;
        if (var < 0.f) var = 0.f;
        const float mean = Feat<FT>::ld(x + c) + ms;
        out_a[c] = mean;
        out_b[c] = 1.0f / sqrtf(var + eps);
        if (run_mean) {
            run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
            const float unb = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
            run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unb;
        }
        if (num_batches && c == 0) *num_batches += 1;
    } else {
        out_a[c] = s2;      // d gamma
        out_b[c] = s1;      // d beta
    }
}

// y = relu?((x - mean) * invstd * gamma + beta)
template <typename FT, typename XT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const XT* __restrict__ x, long long total4, int C,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu,
                                                       FT* __restrict__ y) {
    const int cq = C >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int g = (int)(e % cq);
        const float4 v = Feat<XT>::ld4(x + e * 4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + (g << 2));
        const float4 is = *reinterpret_cast<const float4*>(invstd + (g << 2));
        const float4 ga = *reinterpret_cast<const float4*>(gamma + (g << 2));
        const float4 be = *reinterpret_cast<const float4*>(beta + (g << 2));
        float4 o = make_float4((v.x - mu.x) * is.x * ga.x + be.x, (v.y - mu.y) * is.y * ga.y + be.y,
                               (v.z - mu.z) * is.z * ga.z + be.z, (v.w - mu.w) * is.w * ga.w + be.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        Feat<FT>::st4(y + e * 4, o);
    }
}

// dx = gamma*invstd*(dz - dbeta/R - xhat*dgamma/R)
template <typename FT, typename XT>
__global__ __launch_bounds__(256) void bn_dx_kernel(const XT* __restrict__ x, const FT* __restrict__ dy,
                                                    long long total4, int R, int C, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ dgamma,
                                                    const float* __restrict__ dbeta, int relu, FT* __restrict__ dx,
                                                    int ldy = 0, const FT* __restrict__ dy2 = nullptr, int ldy2 = 0) {
    const int cq = C >> 2;
    const float invR = 1.0f / (float)R;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
        const int g = (int)(e % cq);
        const float4 v = Feat<XT>::ld4(x + e * 4);
        float4 dz;
        if (ldy || dy2) {
            const unsigned row = (unsigned)((unsigned long long)e / (unsigned)cq);          // (32-bit quotient: R < 2^32)
            dz = bn_ld4_pitch<FT>(dy + (size_t)row * (ldy ? ldy : C) + (g << 2), ldy);
            if (dy2) {
                const float4 d2 = bn_ld4_pitch<FT>(dy2 + (size_t)row * ldy2 + (g << 2), ldy2);
                dz.x += d2.x; dz.y += d2.y; dz.z += d2.z; dz.w += d2.w;
            }
        } else {
            dz = Feat<FT>::ld4(dy + e * 4);
        }
        const float4 mu = *reinterpret_cast<const float4*>(mean + (g << 2));
        const float4 is = *reinterpret_cast<const float4*>(invstd + (g << 2));
        const float4 ga = *reinterpret_cast<const float4*>(gamma + (g << 2));
        const float4 be = *reinterpret_cast<const float4*>(beta + (g << 2));
        const float4 dg = *reinterpret_cast<const float4*>(dgamma + (g << 2));
        const float4 db = *reinterpret_cast<const float4*>(dbeta + (g << 2));
        const float4 xh = make_float4((v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w);
        if (relu) {
            if (!(xh.x * ga.x + be.x > 0.f)) dz.x = 0.f;
            if (!(xh.y * ga.y + be.y > 0.f)) dz.y = 0.f;
            if (!(xh.z * ga.z + be.z > 0.f)) dz.z = 0.f;
            if (!(xh.w * ga.w + be.w > 0.f)) dz.w = 0.f;
        }
        float4 o;
        o.x = ga.x * is.x * (dz.x - db.x * invR - xh.x * dg.x * invR);
        o.y = ga.y * is.y * (dz.y - db.y * invR - xh.y * dg.y * invR);
        o.z = ga.z * is.z * (dz.z - db.z * invR - xh.z * dg.z * invR);
        o.w = ga.w * is.w * (dz.w - db.w * invR - xh.w * dg.w * invR);
        Feat<FT>::st4(dx + e * 4, o);
    }
}

static int bn_rows_per_block(int R) {
    int r = (R + BN_MAX_PARTIALS - 1) / BN_MAX_PARTIALS;
    if (r < 32) r = 32;
    return r;
}
static int bn_blocks(int R) { const int r = bn_rows_per_block(R); return (R + r - 1) / r; }

static int bn_check(int R, int C) {
    if (R <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    if ((C & 3) || (BN_THREADS % (C >> 2)) != 0) return HSP_ERR_UNSUPPORTED;   // C in {4,8,...,1024} dividing 1024
    return HSP_OK;
}

static int stream_grid4(long long total4) {
    long long g = (total4 + 255) / 256;
    const long long cap = (long long)HSP_NUM_CU * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace hsp

using namespace hsp;

extern "C" size_t hsp_bn_workspace_bytes(int R, int C) {
    if (R <= 0 || C <= 0) return 0;
    return (size_t)bn_blocks(R) * 2 * C * sizeof(float);
}

template <typename FT, typename XT>
static int bn_relu_fwd_impl(const XT* x, int R, int C, const float* gamma, const float* beta, float eps,
                               float momentum, int relu, FT* y, float* save_mean, float* save_invstd,
                               float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                               size_t ws_bytes, hspStream_t stream) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd) return HSP_ERR_BAD_ARG;
    int rc = bn_check(R, C);
    if (rc) return rc;
    if (!ws || ws_bytes < hsp_bn_workspace_bytes(R, C)) return HSP_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(ws);
    const int nblk = bn_blocks(R);
    hipLaunchKernelGGL((bn_partial_kernel<0, FT, XT>), dim3(nblk), dim3(BN_THREADS), 0, st, x, (const FT*)nullptr, R, C, nullptr, nullptr,
                       nullptr, nullptr, 0, part, bn_rows_per_block(R));
    hipLaunchKernelGGL((bn_finalize_kernel<0, XT>), dim3((C + 15) / 16), dim3(1024), 0, st, part, nblk, R, C, x, eps, momentum,
                       save_mean, save_invstd, running_mean, running_var, num_batches_tracked);
    const long long total4 = (long long)R * (C >> 2);
    hipLaunchKernelGGL((bn_apply_kernel<FT, XT>), dim3(stream_grid4(total4)), dim3(256), 0, st, x, total4, C, save_mean, save_invstd,
                       gamma, beta, relu, y);
    return check_launch();
}

template <typename FT, typename XT>
static int bn_relu_apply_impl(const XT* x, int R, int C, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int relu, FT* y, hspStream_t stream) {
    if (!x || !mean || !invstd || !gamma || !beta || !y) return HSP_ERR_BAD_ARG;
    int rc = bn_check(R, C);
    if (rc) return rc;
    const long long total4 = (long long)R * (C >> 2);
    hipLaunchKernelGGL((bn_apply_kernel<FT, XT>), dim3(stream_grid4(total4)), dim3(256), 0, as_stream(stream), x, total4, C, mean,
                       invstd, gamma, beta, relu, y);
    return check_launch();
}

template <typename FT, typename XT>
static int bn_relu_bwd_impl(const XT* x, const FT* dy, int R, int C, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, int relu, FT* dx, float* dgamma,
                               float* dbeta, void* ws, size_t ws_bytes, hspStream_t stream, int ldy = 0, const FT* dy2 = nullptr,
                               int ldy2 = 0) {
    if (!x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta) return HSP_ERR_BAD_ARG;
    if ((ldy && (ldy < C || (ldy & 1))) || (dy2 && (ldy2 < C || (ldy2 & 1)))) return HSP_ERR_BAD_ARG;     // 8-byte aligned rows
    if ((reinterpret_cast<size_t>(dy) & 7) || (reinterpret_cast<size_t>(dy2) & 7)) return HSP_ERR_BAD_ARG;
    if (sizeof(FT) != 4 && ((ldy & 3) || (ldy2 & 3))) return HSP_ERR_UNSUPPORTED;
    if (ldy == C) ldy = 0;
    int rc = bn_check(R, C);
    if (rc) return rc;
    if (!ws || ws_bytes < hsp_bn_workspace_bytes(R, C)) return HSP_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(ws);
    const int nblk = bn_blocks(R);
    hipLaunchKernelGGL((bn_partial_kernel<1, FT, XT>), dim3(nblk), dim3(BN_THREADS), 0, st, x, dy, R, C, save_mean, save_invstd, gamma,
                       beta, relu, part, bn_rows_per_block(R), ldy, dy2, ldy2);
    hipLaunchKernelGGL((bn_finalize_kernel<1, XT>), dim3((C + 15) / 16), dim3(1024), 0, st, part, nblk, R, C, x, 0.f, 0.f, dgamma,
                       dbeta, nullptr, nullptr, nullptr);
    const long long total4 = (long long)R * (C >> 2);
    hipLaunchKernelGGL((bn_dx_kernel<FT, XT>), dim3(stream_grid4(total4)), dim3(256), 0, st, x, dy, total4, R, C, save_mean,
                       save_invstd, gamma, beta, dgamma, dbeta, relu, dx, ldy, dy2, ldy2);
    return check_launch();
}

extern "C" int hsp_bn_relu_fwd(const float* x, int R, int C, const float* gamma, const float* beta, float eps,
                               float momentum, int relu, float* y, float* save_mean, float* save_invstd,
                               float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                               size_t ws_bytes, hspStream_t stream) {
    return bn_relu_fwd_impl<float, float>(x, R, C, gamma, beta, eps, momentum, relu, y, save_mean, save_invstd, running_mean,
                                   running_var, num_batches_tracked, ws, ws_bytes, stream);
}
/* hsp_bn_relu_fwd whose first pass was done by the producer of x (hsp_gemm_x3_bn_f32): partial[nblk][2][C] shifted sums with
 * the per-column shift `shift` (C floats: any vector known before x is -- the running mean is the natural choice) */
extern "C" int hsp_bn_relu_fwd_partials(const float* x, int R, int C, const float* gamma, const float* beta, float eps,
                                        float momentum, int relu, float* y, float* save_mean, float* save_invstd,
                                        float* running_mean, float* running_var, long long* num_batches_tracked,
                                        const float* partial, int nblk, const float* shift, hspStream_t stream) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !partial || !shift || nblk <= 0 || nblk > BN_MAX_PARTIALS)
        return HSP_ERR_BAD_ARG;
    int rc = bn_check(R, C);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    // (finalize reads its shift through the `x` argument: element c of the first row)
    hipLaunchKernelGGL((bn_finalize_kernel<0, float>), dim3((C + 15) / 16), dim3(1024), 0, st, partial, nblk, R, C, shift, eps, momentum,
                       save_mean, save_invstd, running_mean, running_var, num_batches_tracked);
    const long long total4 = (long long)R * (C >> 2);
    hipLaunchKernelGGL((bn_apply_kernel<float, float>), dim3(stream_grid4(total4)), dim3(256), 0, st, x, total4, C, save_mean, save_invstd,
                       gamma, beta, relu, y);
    return check_launch();
}
extern "C" int hsp_bn_relu_fwd_bf16(const hsp_bf16_t* x, int R, int C, const float* gamma, const float* beta, float eps,
                                    float momentum, int relu, hsp_bf16_t* y, float* save_mean, float* save_invstd,
                                    float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                                    size_t ws_bytes, hspStream_t stream) {
    return bn_relu_fwd_impl<bf16_t, bf16_t>(x, R, C, gamma, beta, eps, momentum, relu, y, save_mean, save_invstd, running_mean,
                                    running_var, num_batches_tracked, ws, ws_bytes, stream);
}
extern "C" int hsp_bn_relu_apply(const float* x, int R, int C, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int relu, float* y, hspStream_t stream) {
    return bn_relu_apply_impl<float, float>(x, R, C, mean, invstd, gamma, beta, relu, y, stream);
}
extern "C" int hsp_bn_relu_apply_bf16(const hsp_bf16_t* x, int R, int C, const float* mean, const float* invstd,
                                      const float* gamma, const float* beta, int relu, hsp_bf16_t* y, hspStream_t stream) {
    return bn_relu_apply_impl<bf16_t, bf16_t>(x, R, C, mean, invstd, gamma, beta, relu, y, stream);
}
extern "C" int hsp_bn_relu_bwd(const float* x, const float* dy, int R, int C, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, int relu, float* dx, float* dgamma,
                               float* dbeta, void* ws, size_t ws_bytes, hspStream_t stream) {
    return bn_relu_bwd_impl<float, float>(x, dy, R, C, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta, ws, ws_bytes,
                                   stream);
}
/* hsp_bn_relu_bwd for a tensor with TWO consumers: dy (rows of pitch ldy >= C elements, a multiple of 4 -- e.g. a column block of
 * a wider gradient tensor, consumed in place) plus an optional second incoming gradient dy2 (pitch ldy2); the kernels add them
 * as they read (the sum autograd would otherwise form with a separate element-wise kernel) */
extern "C" int hsp_bn_relu_bwd2(const float* x, const float* dy, int ldy, const float* dy2, int ldy2, int R, int C,
                                const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, hspStream_t stream) {
    if (ldy <= 0) return HSP_ERR_BAD_ARG;
    return bn_relu_bwd_impl<float, float>(x, dy, R, C, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta, ws, ws_bytes,
                                          stream, ldy, dy2, ldy2);
}
extern "C" int hsp_bn_relu_bwd_bf16(const hsp_bf16_t* x, const hsp_bf16_t* dy, int R, int C, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                    hsp_bf16_t* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                    hspStream_t stream) {
    return bn_relu_bwd_impl<bf16_t, bf16_t>(x, dy, R, C, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta, ws, ws_bytes,
                                    stream);
}

extern "C" int hsp_bn_relu_fwd_mixed(const float* x, int R, int C, const float* gamma, const float* beta, float eps,
                                     float momentum, int relu, hsp_bf16_t* y, float* save_mean, float* save_invstd,
                                     float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                                     size_t ws_bytes, hspStream_t stream) {
    return bn_relu_fwd_impl<bf16_t, float>(x, R, C, gamma, beta, eps, momentum, relu, y, save_mean, save_invstd, running_mean,
                                           running_var, num_batches_tracked, ws, ws_bytes, stream);
}
extern "C" int hsp_bn_relu_apply_mixed(const float* x, int R, int C, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, int relu, hsp_bf16_t* y, hspStream_t stream) {
    return bn_relu_apply_impl<bf16_t, float>(x, R, C, mean, invstd, gamma, beta, relu, y, stream);
}
extern "C" int hsp_bn_relu_bwd_mixed(const float* x, const hsp_bf16_t* dy, int R, int C, const float* gamma,
                                     const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                     hsp_bf16_t* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                     hspStream_t stream) {
    return bn_relu_bwd_impl<bf16_t, float>(x, dy, R, C, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta, ws,
                                           ws_bytes, stream);
}
