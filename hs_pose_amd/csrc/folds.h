// folds.h -- device bodies of the fixed-order folds whose results only the END of a backward pass needs: the split-K
// partials of the parameter gradients (gemm.hip) and the per-cloud partials of the support-direction gradients
// (rfconv.hip).  Each has its own launch (wgrad_reduce*_kernel, rf_dirs_reduce_kernel) and both are callable from ONE
// launch for a whole step (hsp_step_fold, step_fold_kernel): a fold is a 4-8 us dependent launch and an HS stack's backward
// has ten of them, none of which anything before the optimizer reads.  Same summation order in every form: same bits.
#pragma once
#include "common.h"

namespace hsp {

// C[m][n] = sum_s part[s][m][n]; likewise colsum.  256 threads = 64 float4 elements x 4 slice groups: group g sums slices
// g, g+4, ... with loads in flight, the 4 groups are folded through LDS in a fixed order.  blk: block index inside the problem.
__device__ __forceinline__ void wgrad_fold_body(const HspWgradPending& pr, int blk, float4 (*red)[64]) {
    const float* part = reinterpret_cast<const float*>(pr.part);
    const float* cs_part = reinterpret_cast<const float*>(pr.cs_part);
    const int SK = pr.nparts, M = pr.M, N = pr.N;
    const int le = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int nq = N >> 2;
    const long long total = (long long)M * nq;
    const long long ncs = pr.colsum ? (N >> 2) : 0;
    const long long e = (long long)blk * 64 + le;
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
            const int m = (int)(e / nq), qq = (int)(e - (long long)m * nq);
            float* c = reinterpret_cast<float*>(pr.C) + (size_t)m * pr.ldc + (qq << 2);
            c[0] = r.x; c[1] = r.y; c[2] = r.z; c[3] = r.w;    // ldc need not be a multiple of 4
        } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(pr.colsum) + (size_t)(e - total) * 4) = r;
        }
    }
}

// out (3, SC) = J_normalize(dirs) . sum_b ws[b] (3, SC).  64 columns per block x 16 slices of the nblk partials: slice sl sums
// partials sl, sl+32, ... and sl+16, sl+48, ... in two accumulators, the 16 slices are folded through LDS in ascending order,
// then the Jacobian of F.normalize(dim=0) is applied:
//   n = max(||D||, 1e-12);  ||D|| > 1e-12:  gD = (gD^ - D^ (D^ . gD^)) / n ;  else gD = gD^ / 1e-12
// THREADS = 1024: one slice per thread (the stand-alone launch); 256: four slices per thread (inside step_fold_kernel).
// The order of every sum is the same in both, so the two forms agree bit for bit.
template <int THREADS>
__device__ __forceinline__ void dirs_fold_body(const float* __restrict__ ws, int nblk, int SC, const float* __restrict__ dirs,
                                               float* __restrict__ out, int blk, float (*red)[3][64]) {
    const int lj = threadIdx.x & 63;
    const int j = blk * 64 + lj;
    const int n3 = 3 * SC;
#pragma unroll
    for (int sl = threadIdx.x >> 6; sl < 16; sl += THREADS / 64) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        if (j < SC) {
            int b = sl;
            for (; b + 16 < nblk; b += 32) {
                const float* p = ws + (size_t)b * n3 + j;
                const float* q = ws + (size_t)(b + 16) * n3 + j;
                a0 += p[0]; a1 += p[SC]; a2 += p[2 * SC];
                b0 += q[0]; b1 += q[SC]; b2 += q[2 * SC];
            }
            for (; b < nblk; b += 16) {
                const float* p = ws + (size_t)b * n3 + j;
                a0 += p[0]; a1 += p[SC]; a2 += p[2 * SC];
            }
        }
        red[sl][0][lj] = a0 + b0; red[sl][1][lj] = a1 + b1; red[sl][2][lj] = a2 + b2;
    }
    __syncthreads();
    if ((threadIdx.x >> 6) == 0 && j < SC) {
        float g0 = red[0][0][lj], g1 = red[0][1][lj], g2 = red[0][2][lj];
#pragma unroll
        for (int t = 1; t < 16; ++t) { g0 += red[t][0][lj]; g1 += red[t][1][lj]; g2 += red[t][2][lj]; }
        const float x = dirs[j], y = dirs[SC + j], z = dirs[2 * SC + j];
        const float nrm = __fsqrt_rn(add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z)));
        if (nrm > 1e-12f) {
            const float hx = x / nrm, hy = y / nrm, hz = z / nrm;
            const float dot = hx * g0 + hy * g1 + hz * g2;
            out[j] = (g0 - hx * dot) / nrm;
            out[SC + j] = (g1 - hy * dot) / nrm;
            out[2 * SC + j] = (g2 - hz * dot) / nrm;
        } else {
            out[j] = g0 / 1e-12f; out[SC + j] = g1 / 1e-12f; out[2 * SC + j] = g2 / 1e-12f;
        }
    }
}

}  // namespace hsp
