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
#include "folds.h"
#include <stdlib.h>

namespace hsp {

#define RF_THREADS 256

// F.normalize(directions, dim=0) for the 4 columns j..j+3 of the raw (3, S*C) parameter
// (reference gcn3d.py:100,166): D / max(||D||_col, 1e-12).  Folded into every kernel so the normalised
// copy is never materialised and its Jacobian is applied by rf_dirs_reduce_kernel.
__device__ __forceinline__ void load_dirs_normed(const float* __restrict__ dirs, int SC, int j, float4& d0,
                                                 float4& d1, float4& d2) {
    d0 = *reinterpret_cast<const float4*>(dirs + j);
    d1 = *reinterpret_cast<const float4*>(dirs + SC + j);
    d2 = *reinterpret_cast<const float4*>(dirs + 2 * SC + j);
#define RF_NRM(X)                                                                                    \
    {                                                                                                \
        const float n2 = norm2_chain(d0.X, d1.X, d2.X);                                              \
        const float nr = fmaxf(sqrtf(n2), 1e-12f);           /* correctly rounded, as ATen's */          \
        d0.X = __fdiv_rn(d0.X, nr); d1.X = __fdiv_rn(d1.X, nr); d2.X = __fdiv_rn(d2.X, nr);          \
    }
    RF_NRM(x) RF_NRM(y) RF_NRM(z) RF_NRM(w)
#undef RF_NRM
}

// relu(theta) of the forward kernels.  theta = R^ . D^ is the cosine of two unit vectors, so relu == clamp to [0, 1] up to the
// rounding of the three-term chain: fmed3(x, 0, 1) is folded by the compiler into the LAST FMA as its clamp modifier
// (v_fma_f32 ... clamp) -- the 4 v_max of the 41 VALU instructions per (neighbour, float4) unit disappear (the kernel is
// VALU-issue-bound: measured 1.875 -> 1.852 ms per step at B=16 N=1028, 15.92 -> 15.68 ms on the bf16 dense clouds).  The one
// difference from max(x, 0): a theta that rounding pushed an ulp or two ABOVE 1 (a neighbour exactly along a support
// direction) reads 1.0 instead of 1.0000001 (2.4e-7 relative, inside every tolerance of section 2; NaN -> 0 either way).
// -DRF_RELU_MAX restores the plain max.
// RF_WF_POST=1: the pipelined forward fetches the winners' support values AFTER the neighbour loop (four 4-byte gathers per
// float4 unit) instead of carrying them through it as a fourth v_cndmask per element.  Measured in isolation (round 3, same box,
// alternating runs): 1.894 vs 1.874 ms per step at B=16 N=1028, 15.8-16.0 vs 15.63 ms on the bf16 dense clouds -- a wave's four
// scattered 4-byte gathers touch up to 256 cache lines, more address-path work than the 80 selects they replace.  Off.
#ifndef RF_WF_POST
#define RF_WF_POST 0
#endif
#ifdef RF_RELU_MAX
#define RF_RELU(X) fmaxf((X), 0.f)
#else
#define RF_RELU(X) __builtin_amdgcn_fmed3f((X), 0.f, 1.f)
#endif

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
// WF: also record the winners' support values (fwin)
// FT: storage type of fm / out / fwin (float, or bf16_t: arithmetic stays fp32, common.h Feat<>)
template <bool SURFACE, int NCH, bool WF, typename FT>
__global__ __launch_bounds__(RF_THREADS) void rf_fwd_kernel(const float* __restrict__ xyz,
                                                            const int32_t* __restrict__ idx,
                                                            const float* __restrict__ dirs,
                                                            const FT* __restrict__ fm, int B, int N, int k,
                                                            int S, int C, FT* __restrict__ out,
                                                            uint16_t* __restrict__ argrow,
                                                            FT* __restrict__ fwin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int SC = S * C;
    float* smax = reinterpret_cast<float*>(smem);             // SC
    float4* sR = reinterpret_cast<float4*>(smax + SC);        // k  (unit direction, w unused)
    int* sIdx = reinterpret_cast<int*>(sR + k);               // k
    const int tid = threadIdx.x;
    const int nq = SC >> 2;                                   // float4 columns
    const int fstride = (S + 1) * C;
    const float invS_div = (float)S;
    float4 d0[NCH], d1[NCH], d2[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int cq = tid + u * RF_THREADS;
        load_dirs_normed(dirs, SC, (cq < nq ? cq : 0) << 2, d0[u], d1[u], d2[u]);
    }
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
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                const int cq = tid + u * RF_THREADS;
                if (cq < nq) {
                    const int j = cq << 2;
                    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                    int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                    float4 wf = make_float4(1.f, 1.f, 1.f, 1.f);          // the support value behind each winner
                    const FT* fsup = SURFACE ? nullptr : fm + (size_t)b * N * fstride + C + j;
#pragma unroll 4
                    for (int n = 0; n < k; ++n) {
                        const float4 r = sR[n];
                        // theta = relu(R . D) with the k-ordered fma chain of the reference's matmul
                        float4 th;
                        th.x = RF_RELU(__fmaf_rn(r.z, d2[u].x, __fmaf_rn(r.y, d1[u].x, mul_rn(r.x, d0[u].x))));
                        th.y = RF_RELU(__fmaf_rn(r.z, d2[u].y, __fmaf_rn(r.y, d1[u].y, mul_rn(r.x, d0[u].y))));
                        th.z = RF_RELU(__fmaf_rn(r.z, d2[u].z, __fmaf_rn(r.y, d1[u].z, mul_rn(r.x, d0[u].z))));
                        th.w = RF_RELU(__fmaf_rn(r.z, d2[u].w, __fmaf_rn(r.y, d1[u].w, mul_rn(r.x, d0[u].w))));
                        if (!SURFACE) {
                            const float4 f = Feat<FT>::ld4(fsup + (size_t)sIdx[n] * fstride);
                            th.x = mul_rn(th.x, f.x); th.y = mul_rn(th.y, f.y);
                            th.z = mul_rn(th.z, f.z); th.w = mul_rn(th.w, f.w);
                            if (WF) {
                                if (th.x > best.x) wf.x = f.x;
                                if (th.y > best.y) wf.y = f.y;
                                if (th.z > best.z) wf.z = f.z;
                                if (th.w > best.w) wf.w = f.w;
                            }
                        }
                        if (th.x > best.x) { best.x = th.x; a0 = n; }
                        if (th.y > best.y) { best.y = th.y; a1 = n; }
                        if (th.z > best.z) { best.z = th.z; a2 = n; }
                        if (th.w > best.w) { best.w = th.w; a3 = n; }
                    }
                    // ... and that winner's support value fm[b,m*,C+j]: the backward's direction gradient reads it
                    // as a stream instead of gathering 4-byte values from N rows (4x the HBM traffic, measured)
                    // fwin / argrow / out are write-once streams read again only by the backward: non-temporal stores keep
                    // them from evicting the cloud's fm rows (the gather's working set) from the XCD's L2
                    if (WF) Feat<FT>::st4_nt(fwin + pt * SC + j, wf);
                    *reinterpret_cast<float4*>(smax + j) = best;
                    // the winning SOURCE ROW m* = idx[b,i,n*] (uint16): the backward needs neither idx nor n
                    {
                        const unsigned lo = (unsigned)sIdx[a0] | ((unsigned)sIdx[a1] << 16);
                        const unsigned hi = (unsigned)sIdx[a2] | ((unsigned)sIdx[a3] << 16);
                        unsigned* ap = reinterpret_cast<unsigned*>(argrow + pt * SC + j);
                        __builtin_nontemporal_store(lo, ap); __builtin_nontemporal_store(hi, ap + 1);
                    }
                }
            }
            __syncthreads();
            for (int c = tid; c < C; c += RF_THREADS) {
                float s = smax[c];
                for (int sp = 1; sp < S; ++sp) s = add_rn(s, smax[sp * C + c]);
                float v = __fdiv_rn(s, invS_div);
                if (!SURFACE) v = add_rn(Feat<FT>::ld(fm + pt * fstride + c), v);
                Feat<FT>::st_nt(out + pt * C + c, v);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// forward, PIPELINED over the points of a workgroup (the default schedule).  rf_fwd_kernel spends a point in three phases
// separated by barriers: k threads fetch the neighbour list and build the unit directions (two dependent global round
// trips while 236 threads wait), everybody gathers, C threads average the S maxima.  Here the three phases of three
// consecutive points share ONE barrier interval: the threads that have no column group in the last of the NCH slots
// (S*C/4 = 224 groups on 256 threads at C = 128) build the directions of point i+1 into the other half of a double
// buffer while the gather of point i runs, and the average of point i-1 is taken from the other maxima buffer at the
// start of the interval.  Same arithmetic, same results bit for bit (tests/test_gpu_layers.py compares the schedules).
// dynamic LDS: 2 * (S*C + 6 k) floats.  Needs S*C/4 - (NCH-1)*256 + k <= 256.
// ------------------------------------------------------------------------------------------------
template <bool SURFACE, int NCH, bool WF, typename FT>
__global__ __launch_bounds__(RF_THREADS) void rf_fwd_pipe_kernel(const float* __restrict__ xyz,
                                                                 const int32_t* __restrict__ idx,
                                                                 const float* __restrict__ dirs,
                                                                 const FT* __restrict__ fm, int B, int N, int k,
                                                                 int S, int C, FT* __restrict__ out,
                                                                 uint16_t* __restrict__ argrow,
                                                                 FT* __restrict__ fwin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int SC = S * C;
    float* smax2 = reinterpret_cast<float*>(smem);            // 2 x SC
    float4* sR2 = reinterpret_cast<float4*>(smax2 + 2 * SC);  // 2 x k
    int* sIdx2 = reinterpret_cast<int*>(sR2 + 2 * k);         // 2 x k
    unsigned* sOff2 = reinterpret_cast<unsigned*>(sIdx2 + 2 * k);   // 2 x k: the neighbour row's byte offset in the cloud's fm
    const int tid = threadIdx.x;
    const int nq = SC >> 2;
    const int fstride = (S + 1) * C;
    const float invS_div = (float)S;
    const int spare0 = nq - (NCH - 1) * RF_THREADS;           // first thread without a group in the last slot
    float4 d0[NCH], d1[NCH], d2[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int cq = tid + u * RF_THREADS;
        load_dirs_normed(dirs, SC, (cq < nq ? cq : 0) << 2, d0[u], d1[u], d2[u]);
    }
    const PointIter it(B);
    auto dirs_phase = [&](int b, int i, int buf) {
        const int t = tid - spare0;
        if (t >= 0 && t < k) {
            const float* xb = xyz + (size_t)b * N * 3;
            const int m = idx[((size_t)b * N + i) * k + t];
            sIdx2[buf * k + t] = m;
            sOff2[buf * k + t] = (unsigned)m * (unsigned)fstride * (unsigned)sizeof(FT);
            const float3 r = unit_dir(xb[i * 3], xb[i * 3 + 1], xb[i * 3 + 2], xb[m * 3], xb[m * 3 + 1], xb[m * 3 + 2]);
            sR2[buf * k + t] = make_float4(r.x, r.y, r.z, 0.f);
        }
    };
    auto mean_phase = [&](size_t pt, const float* smax) {
        for (int c = tid; c < C; c += RF_THREADS) {
            float s = smax[c];
            for (int sp = 1; sp < S; ++sp) s = add_rn(s, smax[sp * C + c]);
            float v = __fdiv_rn(s, invS_div);
            if (!SURFACE) v = add_rn(Feat<FT>::ld(fm + pt * fstride + c), v);
            Feat<FT>::st_nt(out + pt * C + c, v);
        }
    };
    int b = it.b0, i = it.i0;
    bool have = b < B && i < N;
    if (have) dirs_phase(b, i, 0);
    int cur = 0;
    bool have_prev = false;
    size_t prev_pt = 0;
    while (have) {
        int nb = b, ni = i + it.istep;
        if (ni >= N) { nb = b + it.bstep; ni = it.i0; }
        const bool have_next = nb < B;
        const size_t pt = (size_t)b * N + i;
        __syncthreads();        // directions of this point are in place; the previous point's maxima are complete
        if (have_prev) mean_phase(prev_pt, smax2 + (cur ^ 1) * SC);
        if (have_next) dirs_phase(nb, ni, cur ^ 1);
        float* smax = smax2 + cur * SC;
        const float4* sR = sR2 + cur * k;
        const int* sIdx = sIdx2 + cur * k;
        const unsigned* sOff = sOff2 + cur * k;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int cq = tid + u * RF_THREADS;
            if (cq < nq) {
                const int j = cq << 2;
                float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                float4 wf = make_float4(1.f, 1.f, 1.f, 1.f);
                // gathers through a buffer descriptor over the cloud's fm: address = base + 32-bit (column + row) byte offset, one
                // v_add per gather instead of a 64-bit multiply-add
                // (the base goes through v_readfirstlane: hipcc does not see that the cloud index is wave-uniform and would wrap
                // every load in a waterfall loop over the descriptor)
                const unsigned long long fbase = reinterpret_cast<unsigned long long>(SURFACE ? out : fm + (size_t)b * N * fstride);
                FT* ubase = reinterpret_cast<FT*>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(fbase >> 32)) << 32) |
                                                  (unsigned)__builtin_amdgcn_readfirstlane((unsigned)fbase));
                const __amdgpu_buffer_rsrc_t frs =
                    __builtin_amdgcn_make_buffer_rsrc(ubase, 0, (int)((size_t)N * fstride * sizeof(FT)), 0x00020000);
                const unsigned voff = (unsigned)(C + j) * (unsigned)sizeof(FT);
#pragma unroll 4
                for (int n = 0; n < k; ++n) {
                    const float4 r = sR[n];
                    float4 th;
                    th.x = RF_RELU(__fmaf_rn(r.z, d2[u].x, __fmaf_rn(r.y, d1[u].x, mul_rn(r.x, d0[u].x))));
                    th.y = RF_RELU(__fmaf_rn(r.z, d2[u].y, __fmaf_rn(r.y, d1[u].y, mul_rn(r.x, d0[u].y))));
                    th.z = RF_RELU(__fmaf_rn(r.z, d2[u].z, __fmaf_rn(r.y, d1[u].z, mul_rn(r.x, d0[u].z))));
                    th.w = RF_RELU(__fmaf_rn(r.z, d2[u].w, __fmaf_rn(r.y, d1[u].w, mul_rn(r.x, d0[u].w))));
                    if (!SURFACE) {
                        float4 f;
                        if constexpr (sizeof(FT) == 4) {
                            const auto v = __builtin_amdgcn_raw_buffer_load_b128(frs, (int)(voff + sOff[n]), 0, 0);
                            f = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
                        } else {
                            const auto v = __builtin_amdgcn_raw_buffer_load_b64(frs, (int)(voff + sOff[n]), 0, 0);
                            f = make_float4(__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u),
                                            __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u));
                        }
                        th.x = mul_rn(th.x, f.x); th.y = mul_rn(th.y, f.y);
                        th.z = mul_rn(th.z, f.z); th.w = mul_rn(th.w, f.w);
                        if (WF && !RF_WF_POST) {
                            if (th.x > best.x) wf.x = f.x;
                            if (th.y > best.y) wf.y = f.y;
                            if (th.z > best.z) wf.z = f.z;
                            if (th.w > best.w) wf.w = f.w;
                        }
                    }
                    if (th.x > best.x) { best.x = th.x; a0 = n; }
                    if (th.y > best.y) { best.y = th.y; a1 = n; }
                    if (th.z > best.z) { best.z = th.z; a2 = n; }
                    if (th.w > best.w) { best.w = th.w; a3 = n; }
                }
                if (WF && RF_WF_POST && !SURFACE) {
                    // (opt-in, measured slower: see RF_WF_POST above)
                    constexpr unsigned ES = (unsigned)sizeof(FT);
                    const unsigned o0 = voff + sOff[a0], o1 = voff + sOff[a1] + ES, o2 = voff + sOff[a2] + 2 * ES,
                                   o3 = voff + sOff[a3] + 3 * ES;
                    if constexpr (sizeof(FT) == 4) {
                        wf.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(frs, (int)o0, 0, 0));
                        wf.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(frs, (int)o1, 0, 0));
                        wf.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(frs, (int)o2, 0, 0));
                        wf.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(frs, (int)o3, 0, 0));
                    } else {
                        wf.x = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(frs, (int)o0, 0, 0) << 16);
                        wf.y = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(frs, (int)o1, 0, 0) << 16);
                        wf.z = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(frs, (int)o2, 0, 0) << 16);
                        wf.w = __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(frs, (int)o3, 0, 0) << 16);
                    }
                }
                *reinterpret_cast<float4*>(smax + j) = best;
                {
                    const unsigned lo = (unsigned)sIdx[a0] | ((unsigned)sIdx[a1] << 16);
                    const unsigned hi = (unsigned)sIdx[a2] | ((unsigned)sIdx[a3] << 16);
                    unsigned* ap = reinterpret_cast<unsigned*>(argrow + pt * SC + j);
                    __builtin_nontemporal_store(lo, ap); __builtin_nontemporal_store(hi, ap + 1);
                }
                if (WF) Feat<FT>::st4_nt(fwin + pt * SC + j, wf);
            }
        }
        have_prev = true; prev_pt = pt;
        cur ^= 1;
        b = nb; i = ni; have = have_next;
    }
    __syncthreads();
    if (have_prev) mean_phase(prev_pt, smax2 + (cur ^ 1) * SC);
}

// ------------------------------------------------------------------------------------------------
// backward of HS_layer.graph_conv, GATHER form (no atomics, fixed summation order).
// A workgroup owns one SOURCE row m at a time and walks the reverse-edge list of m (csr.hip):
// for every edge e = i*k + n with idx[b,i,n] == m and every column j
//     hit = (argrow[b,i,j] == m);  z = R(i->m) . dirs[:,j];  ga = g[b,i,j%C] / S
//     grad_fm[b,m,C+j] += hit ? ga * relu(z) : 0
//     gD[d][j]         += hit && z>0 ? ga * fm[b,m,C+j] * R[d] : 0        (registers -> ws partials)
// grad_fm[b,m,c] = g[b,m,c] (centre).  Edges are staged EB at a time in LDS (unit direction, slot n,
// the query's g row pre-scaled by 1/S) so the EB arg-max loads of a thread are independent.
// dynamic LDS: EB*(C + 8) floats
// ------------------------------------------------------------------------------------------------
#define RF_EB 8

template <int NCH>
__global__ __launch_bounds__(RF_THREADS) void rf_conv_bwd_csr_kernel(
    const float* __restrict__ xyz, const float* __restrict__ dirs, const float* __restrict__ fm,
    const uint16_t* __restrict__ argrow, const float* __restrict__ gout, const int32_t* __restrict__ rev_off,
    const int32_t* __restrict__ rev_edge, int B, int N, int k, int S, int C, float* __restrict__ gfm,
    float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int SC = S * C;
    float* sg = reinterpret_cast<float*>(smem);                       // RF_EB x C   (g[i] / S)
    float4* sR = reinterpret_cast<float4*>(sg + RF_EB * C);           // RF_EB       (R.xyz, w = slot n as int bits)
    int* sI = reinterpret_cast<int*>(sR + RF_EB);                     // RF_EB       query row i
    const int tid = threadIdx.x;
    const int nq = SC >> 2;
    const int fstride = (S + 1) * C;
    const float invS = 1.0f / (float)S;
    const int cq4 = C >> 2;

    float4 d0[NCH], d1[NCH], d2[NCH], gd0[NCH], gd1[NCH], gd2[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        const int cq = tid + u * RF_THREADS;
        const int j = (cq < nq ? cq : 0) << 2;
        load_dirs_normed(dirs, SC, j, d0[u], d1[u], d2[u]);
        gd0[u] = gd1[u] = gd2[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const PointIter it(B);
    for (int b = it.b0; b < B; b += it.bstep) {
        const float* xb = xyz + (size_t)b * N * 3;
        const int32_t* offb = rev_off + (size_t)b * (N + 1);
        const int32_t* edgeb = rev_edge + (size_t)b * N * k;
        for (int m = it.i0; m < N; m += it.istep) {
            const size_t pm = (size_t)b * N + m;
            const int o0 = offb[m], o1 = offb[m + 1];
            const float mx = xb[m * 3], my = xb[m * 3 + 1], mz = xb[m * 3 + 2];
            float4 fv[NCH], acc[NCH];
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                const int cq = tid + u * RF_THREADS;
                acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                fv[u] = cq < nq ? *reinterpret_cast<const float4*>(fm + pm * fstride + C + (cq << 2)) : acc[u];
            }
            for (int c = tid; c < C; c += RF_THREADS) gfm[pm * fstride + c] = gout[pm * C + c];   // centre
            for (int e0 = o0; e0 < o1; e0 += RF_EB) {
                const int ne = min(RF_EB, o1 - e0);
                __syncthreads();                                   // previous batch consumed
                if (tid < ne) {
                    const int e = edgeb[e0 + tid];
                    const int i = e / k, n = e - i * k;
                    const float3 r = unit_dir(xb[i * 3], xb[i * 3 + 1], xb[i * 3 + 2], mx, my, mz);
                    sR[tid] = make_float4(r.x, r.y, r.z, __int_as_float(n));
                    sI[tid] = i;
                }
                __syncthreads();
                for (int q = tid; q < ne * cq4; q += RF_THREADS) {   // stage the g rows, scaled by 1/S
                    const int t = q / cq4, c4 = q - t * cq4;
                    float4 g = *reinterpret_cast<const float4*>(gout + ((size_t)b * N + sI[t]) * C + (c4 << 2));
                    g.x *= invS; g.y *= invS; g.z *= invS; g.w *= invS;
                    *reinterpret_cast<float4*>(sg + t * C + (c4 << 2)) = g;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int cq = tid + u * RF_THREADS;
                    if (cq < nq) {
                        const int j = cq << 2;
                        const int c = j % C;
                        ushort4 am[RF_EB];
#pragma unroll
                        for (int t = 0; t < RF_EB; ++t)     // clamped slot instead of a branch around the load
                            am[t] = *reinterpret_cast<const ushort4*>(argrow + ((size_t)b * N + sI[t < ne ? t : 0]) * SC + j);
#pragma unroll
                        for (int t = 0; t < RF_EB; ++t) {
                            if (t < ne) {
                                const float4 r = sR[t];
                                const int n = m;                    // a hit is "the winner of (i,j) is this row"
                                const float4 ga = *reinterpret_cast<const float4*>(sg + t * C + c);
#define RF_ONE(X)                                                                                         \
    if (am[t].X == n) {                                                                                   \
        const float z = __fmaf_rn(r.z, d2[u].X, __fmaf_rn(r.y, d1[u].X, mul_rn(r.x, d0[u].X)));           \
        if (z > 0.f) {                                                                                    \
            acc[u].X += ga.X * z;                                                                         \
            const float w = ga.X * fv[u].X;                                                               \
            gd0[u].X += w * r.x; gd1[u].X += w * r.y; gd2[u].X += w * r.z;                                \
        }                                                                                                 \
    }
                                RF_ONE(x) RF_ONE(y) RF_ONE(z) RF_ONE(w)
#undef RF_ONE
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                const int cq = tid + u * RF_THREADS;
                if (cq < nq) *reinterpret_cast<float4*>(gfm + pm * fstride + C + (cq << 2)) = acc[u];
            }
        }
    }
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

// ------------------------------------------------------------------------------------------------
// backward, COLUMN-TILE LDS-SCATTER form (default for HS_layer.graph_conv; SURFACE = the surface layer,
// which has only the direction gradient).
// One workgroup per (cloud b, tile of TC support columns).  The tile of grad_fm, acc[N][TC] fp32, and the
// cloud's xyz live in LDS (66 KB + 12 KB at N=1028, TC=16); the workgroup sweeps the cloud's points i and,
// for each column j of the tile, reads the winning source row m = argrow[b,i,j] (coalesced), rebuilds
// R(i->m) from the LDS copy of xyz, and routes  ga*relu(z)  to acc[m] with an LDS atomic add
// (ds_add_f32: no L2 round trip; hub rows of feature-space KNN graphs only serialise inside one wave);
// every grad_fm row segment is then written once with 16-byte stores.  The fm[b,m,C+j] value per element
// that the direction gradient needs comes from fwin[b,i,j], stored by the forward next to argrow (a stream,
// where gathering it from fm fetched a 64-byte sector per 4-byte value).  That gradient is
// accumulated in registers, folded across the workgroup's point lanes in LDS in a fixed order and written
// to gd_part[b][3][SC] (one writer per element); rf_dirs_reduce_kernel sums the B clouds.
// (hsp_rf_conv_bwd, the CSR form, is the gather twin; both are bit-reproducible now that the tile adds integers.)
// grid (SC/TC, B), block 512, dynamic LDS = max(N*TC (acc) + 3N (xyz), 512*12) floats
// ------------------------------------------------------------------------------------------------
#ifndef RF_BWD_CLOUD_PER_XCD
#define RF_BWD_CLOUD_PER_XCD 1     // 0: only groups of 4 adjacent tiles share an XCD (measured 65.3 vs 60.7 us at N = 1028)
#endif
#define RF_TILE_THREADS 512   // 8 waves: with one 78 KB tile per workgroup this doubles the waves per CU
// The tile accumulates in 32-bit FIXED POINT: ds_add_u32 retires ~24x the lanes per clock of ds_add_f32 on gfx950
// (tools/ubench/lds_atomic.hip), and with the winners' support values streamed (fwin) the float adds were what the
// cache-resident layers waited on (measured: 52.7 -> 28.1 us at N = 257, C = 256; 83 -> 72 us at N = 1028).  The scale is
// private to the workgroup -- a power of two chosen from the largest |grad_out| / S of the columns this tile reads, with
// ceil(log2 N) bits of headroom: a (row, column) cell receives at most one term per point and every term is bounded by that
// maximum (|theta| <= 1), so the sum cannot overflow; a term is rounded to 2^-(30 - log2 N) of the tile's largest one (1.2e-7
// at N = 257, 1e-6 at N = 1028 -- the fp32 adds it replaces carried the same order) and the integer sum is exact and
// order-independent: the scatter form is now bit-reproducible as well.
#define RF_ACC_ADD(P, V) atomicAdd(reinterpret_cast<int*>(P), __float2int_rn((V) * fx_scale))

// FWIN: the support values come from the forward's fwin stream; else they are gathered from fm (fine while a
// cloud's fm stays L2-resident: small N)
// RS > 1 (dense clouds: N = 4096): the tile holds only the source rows [r0, r0 + N/RS) of the cloud -- RS workgroups (grid z)
// sweep the same points and each keeps the elements whose winning row falls in its range.  What LDS buys is bytes of
// accumulator: lines touched per launch = N * SC * RS / TC, and acc = (N / RS) * TC * 4 bytes, so two 128 KB half-cloud
// tiles of 16 columns touch half the lines of one 64 KB whole-cloud tile of 4 (the only whole-cloud width that fits).
template <int TC, bool SURFACE, bool FWIN, typename FT, int RS = 1>
__global__ __launch_bounds__(RF_TILE_THREADS) void rf_bwd_tile_kernel(
    const float* __restrict__ xyz, const float* __restrict__ dirs, const FT* __restrict__ fwin,
    const uint16_t* __restrict__ argrow, const FT* __restrict__ gout, int B, int N, int S, int C,
    FT* __restrict__ gfm, float* __restrict__ gd_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NR = RS > 1 ? (N + RS - 1) / RS : N;                     // source rows of this tile
    const int r0 = RS > 1 ? (int)blockIdx.z * NR : 0;
    const int r1 = min(N, r0 + NR);
    float* acc = reinterpret_cast<float*>(smem);                       // NR*TC   (unused when SURFACE)
    float* sx = acc + (SURFACE ? 0 : (size_t)NR * TC);                 // 3*NR: xyz of the source rows (RS == 1: of all points)
    constexpr int G = TC / 4;                 // float4 groups per tile
    constexpr int PL = RF_TILE_THREADS / G;        // point lanes
    const int SC = S * C;
    const int fstride = (S + 1) * C;
    const int tid = threadIdx.x;
    // (cloud, column tile) of this workgroup.  Neighbouring column tiles read neighbouring 32- / 64-byte pieces of the same
    // argrow / fwin / grad_out lines; dispatched round-robin over the 8 XCDs each piece was fetched into another L2 (measured:
    // 340 MB of HBM traffic for 164 MB algorithmic at N = 1028).  Groups of 4 adjacent tiles are therefore given to ONE XCD, as
    // consecutive workgroups of it (block id = XCD + 8 k, observed placement: a speed choice only).
    int b = blockIdx.y, tile = blockIdx.x;
    {
        const int T = gridDim.x, ngrp = (T >> 2) * (int)gridDim.y;
            if (RF_BWD_CLOUD_PER_XCD && ((int)gridDim.y & 7) == 0) {     // every tile of a cloud on one XCD (grad_out is shared by its S tiles too)
            const int L = blockIdx.x + T * blockIdx.y;
            const int xcd = L & 7, k = L >> 3;
            b = xcd + 8 * (k / T);
            tile = k - (k / T) * T;
        } else if ((T & 3) == 0 && (ngrp & 7) == 0) {
            const int L = blockIdx.x + T * blockIdx.y;
            const int xcd = L & 7, k = L >> 3;
            const int gid = xcd + 8 * (k >> 2);
            b = gid / (T >> 2);
            tile = 4 * (gid - b * (T >> 2)) + (k & 3);
        }
    }
    const int j0 = tile * TC;
    const int cg = tid % G, pl = tid / G;
    const int j = j0 + cg * 4;
    const int c = j % C;
    const float invS = 1.0f / (float)S;
    const float* xb = xyz + (size_t)b * N * 3;
    if (!SURFACE)
        for (int q = tid; q < NR * G; q += RF_TILE_THREADS) *reinterpret_cast<float4*>(acc + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = tid; q < 3 * (r1 - r0); q += RF_TILE_THREADS) sx[q] = xb[r0 * 3 + q];
    // fixed-point scale of this tile: max |grad_out| / S over the (point, channel) values it is going to route
    float fx_scale = 1.f, fx_inv = 1.f;
    if (!SURFACE) {
        __shared__ float wmax[RF_TILE_THREADS / 64];
        float vm = 0.f;
        for (int p = pl; p < N; p += PL) {
            const float4 gq = Feat<FT>::ld4(gout + ((size_t)b * N + p) * C + c);
            vm = fmaxf(vm, fmaxf(fmaxf(fabsf(gq.x), fabsf(gq.y)), fmaxf(fabsf(gq.z), fabsf(gq.w))));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vm = fmaxf(vm, __shfl_xor(vm, o));
        if ((tid & 63) == 0) wmax[tid >> 6] = vm;
        __syncthreads();
        vm = wmax[0];
#pragma unroll
        for (int w = 1; w < RF_TILE_THREADS / 64; ++w) vm = fmaxf(vm, wmax[w]);
        vm *= invS;
        if (vm > 0.f && vm < 3.0e38f) {            // (0 / inf / NaN gradients: scale 1 -- nothing sensible to preserve)
            int ex;
            frexpf(vm, &ex);                        // vm < 2^ex
            int lg = 0;
            while ((1 << lg) < N) ++lg;
            int e = 30 - ex - lg;
            e = e > 120 ? 120 : (e < -120 ? -120 : e);
            fx_scale = ldexpf(1.f, e);
            fx_inv = ldexpf(1.f, -e);
        }
    }
    float4 d0, d1, d2;
    load_dirs_normed(dirs, SC, j, d0, d1, d2);
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
    const FT* fsup = (SURFACE || FWIN) ? nullptr : fwin + (size_t)b * N * fstride + C + j;   // (fwin is fm then)
    __syncthreads();
    // software pipeline: the next point's winning rows / their support values / gradient are in flight while
    // this one is processed; with FWIN every global read of the loop is a coalesced stream
    ushort4 am_n = make_ushort4(0, 0, 0, 0);
    float4 ga_n = make_float4(0.f, 0.f, 0.f, 0.f), fw_n = make_float4(1.f, 1.f, 1.f, 1.f);
    if (pl < N) {
        am_n = *reinterpret_cast<const ushort4*>(argrow + ((size_t)b * N + pl) * SC + j);
        ga_n = Feat<FT>::ld4(gout + ((size_t)b * N + pl) * C + c);
        if (!SURFACE && FWIN) fw_n = Feat<FT>::ld4(fwin + ((size_t)b * N + pl) * SC + j);
    }
    for (int p = pl; p < N; p += PL) {
        const ushort4 am = am_n;
        float4 ga = ga_n;
        const float4 fw = fw_n;
        const int pn = p + PL < N ? p + PL : p;                      // clamped: no branch around the loads
        am_n = *reinterpret_cast<const ushort4*>(argrow + ((size_t)b * N + pn) * SC + j);
        ga_n = Feat<FT>::ld4(gout + ((size_t)b * N + pn) * C + c);
        if (!SURFACE && FWIN) fw_n = Feat<FT>::ld4(fwin + ((size_t)b * N + pn) * SC + j);
        float f0 = fw.x, f1 = fw.y, f2 = fw.z, f3 = fw.w;
        if (!SURFACE && !FWIN) {
            f0 = Feat<FT>::ld(fsup + (size_t)am.x * fstride + 0);
            f1 = Feat<FT>::ld(fsup + (size_t)am.y * fstride + 1);
            f2 = Feat<FT>::ld(fsup + (size_t)am.z * fstride + 2);
            f3 = Feat<FT>::ld(fsup + (size_t)am.w * fstride + 3);
        }
        ga.x *= invS; ga.y *= invS; ga.z *= invS; ga.w *= invS;
        float px, py, pz;
        if (RS > 1) { px = xb[p * 3]; py = xb[p * 3 + 1]; pz = xb[p * 3 + 2]; }      // the point itself: a coalesced stream
        else { px = sx[p * 3]; py = sx[p * 3 + 1]; pz = sx[p * 3 + 2]; }
#define RF_T1(X, E, FV)                                                                              \
        {                                                                                            \
            const int m = (int)am.X - r0;                                                            \
            if (RS == 1 || (unsigned)m < (unsigned)(r1 - r0)) {                                      \
                const float3 r = unit_dir_fast(px, py, pz, sx[m * 3], sx[m * 3 + 1], sx[m * 3 + 2]); \
                const float z = __fmaf_rn(r.z, d2.X, __fmaf_rn(r.y, d1.X, mul_rn(r.x, d0.X)));       \
                if (z > 0.f) {                                                                       \
                    if (!SURFACE) RF_ACC_ADD(acc + m * TC + cg * 4 + E, ga.X * z);                   \
                    const float w = ga.X * FV;                                                       \
                    g0.X += w * r.x; g1.X += w * r.y; g2.X += w * r.z;                               \
                }                                                                                    \
            }                                                                                        \
        }
        RF_T1(x, 0, f0) RF_T1(y, 1, f1) RF_T1(z, 2, f2) RF_T1(w, 3, f3)
#undef RF_T1
    }
    __syncthreads();
    if (!SURFACE) {
        // flush the tile: one 16-byte store per (row, group)
        for (int q = tid; q < (r1 - r0) * G; q += RF_TILE_THREADS) {
            const int m = q / G, g4 = q - m * G;
            const int4 iv = *reinterpret_cast<const int4*>(acc + m * TC + g4 * 4);
            Feat<FT>::st4(gfm + ((size_t)b * N + r0 + m) * fstride + C + j0 + g4 * 4,
                          make_float4((float)iv.x * fx_inv, (float)iv.y * fx_inv, (float)iv.z * fx_inv, (float)iv.w * fx_inv));
        }
        if (j0 < C) {   // the first C/TC tiles also copy the centre columns grad_fm[b,m,c] = g[b,m,c]
            for (int q = tid; q < (r1 - r0) * G; q += RF_TILE_THREADS) {
                const int m = q / G, g4 = q - m * G;
                Feat<FT>::st4(gfm + ((size_t)b * N + r0 + m) * fstride + j0 + g4 * 4,
                              Feat<FT>::ld4(gout + ((size_t)b * N + r0 + m) * C + j0 + g4 * 4));
            }
        }
        __syncthreads();
    }
    // fold the direction gradient across the PL point lanes (fixed order) -> gd_part[b][d][j]
    float* red = reinterpret_cast<float*>(smem);       // RF_TILE_THREADS x 12 floats
    float* mine = red + tid * 12;
    mine[0] = g0.x; mine[1] = g0.y; mine[2] = g0.z; mine[3] = g0.w;
    mine[4] = g1.x; mine[5] = g1.y; mine[6] = g1.z; mine[7] = g1.w;
    mine[8] = g2.x; mine[9] = g2.y; mine[10] = g2.z; mine[11] = g2.w;
    __syncthreads();
    if (tid < G * 12) {
        const int gg = tid / 12, w = tid - gg * 12;    // group, (d*4 + e)
        float sacc = 0.f;
        for (int l = 0; l < PL; ++l) sacc += red[(l * G + gg) * 12 + w];
        const int d = w >> 2, e = w & 3;
        gd_part[((size_t)(b * RS + (RS > 1 ? (int)blockIdx.z : 0)) * 3 + d) * SC + j0 + gg * 4 + e] = sacc;
    }
}

// Direction-gradient epilogue.  ws[blk][3][SC] holds per-workgroup partials of d(loss)/d(D^) (D^ = the
// column-normalised directions).  Workgroup = 64 columns x 16 block-slices: every thread sums its slice
// for the 3 rows of its column (independent accumulators keep loads in flight), the 16 slices are folded
// through LDS in a fixed order (deterministic), then the Jacobian of F.normalize(dim=0) is applied:
//   n = max(||D||, 1e-12);  ||D|| > 1e-12:  gD = (gD^ - D^ (D^ . gD^)) / n ;  else gD = gD^ / 1e-12
__global__ __launch_bounds__(1024) void rf_dirs_reduce_kernel(const float* __restrict__ ws, int nblk, int SC,
                                                              const float* __restrict__ dirs,
                                                              float* __restrict__ out) {
    __shared__ float red[16][3][64];
    dirs_fold_body<1024>(ws, nblk, SC, dirs, out, (int)blockIdx.x, red);          // (folds.h; also inside hsp_step_fold)
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
    if (N > 65535 || (C & 3)) return HSP_ERR_UNSUPPORTED;   // winning rows are stored as uint16; float4 column groups
    return HSP_OK;
}

template <bool SURFACE, typename FT>
static int rf_fwd(const float* xyz, const int32_t* idx, const float* dirs, const FT* fm, int B, int N, int k,
                  int S, int C, FT* out, uint16_t* argrow, FT* fwin, hspStream_t stream) {
    int rc = rf_check(xyz, idx, dirs, B, N, k, S, C);
    if (rc) return rc;
    if (!out || !argrow || (!SURFACE && !fm)) return HSP_ERR_BAD_ARG;
    // (A channel-split schedule -- C / 2 channels per pass so that a cloud's fm slice stays in the XCD's L2 -- halves the fabric
    // reads (FETCH_SIZE 252 -> 121 MB at B=16 N=1028) but the kernel is VALU-issue-bound: 121 vs 101 us in round 2, and again
    // slower inside the matrix-core form of round 4, tools/experiments/.  Removed.)
    const size_t lds = (size_t)(S * C + 5 * k) * 4;
    if (lds > 64 * 1024) return HSP_ERR_UNSUPPORTED;
    // workgroups per CU: 8 for the large layers; small clouds amortise a workgroup's prologue (direction loads + normalisation)
    // over more points with 6 / 4 (measured at B = 16: N = 257 55.3 -> 52.4 us, N = 64 24.3 -> 23.6 us)
    const long long npts = (long long)B * N;
    const int grid = persistent_blocks(npts, npts < 2048 ? 4 : npts < 8192 ? 6 : 8);
    const int nch = ((S * C >> 2) + RF_THREADS - 1) / RF_THREADS;
    if (nch > 4) return HSP_ERR_UNSUPPORTED;              // S*C <= 4096
    {
        // the pipelined schedule (three points' phases per barrier interval) wherever the last slot leaves k threads free
        const int spare0 = (S * C >> 2) - (nch - 1) * RF_THREADS;
        const size_t lds_pipe = 2 * (size_t)(S * C + 6 * k) * 4;
        if (spare0 + k <= RF_THREADS && lds_pipe <= 64 * 1024) {
#define RF_PIPE_LAUNCH(NCH)                                                                                        \
    if (!SURFACE && fwin)                                                                                          \
        hipLaunchKernelGGL((rf_fwd_pipe_kernel<SURFACE, NCH, !SURFACE, FT>), dim3(grid), dim3(RF_THREADS), lds_pipe, \
                           as_stream(stream), xyz, idx, dirs, fm, B, N, k, S, C, out, argrow, fwin);               \
    else                                                                                                           \
        hipLaunchKernelGGL((rf_fwd_pipe_kernel<SURFACE, NCH, false, FT>), dim3(grid), dim3(RF_THREADS), lds_pipe,   \
                           as_stream(stream), xyz, idx, dirs, fm, B, N, k, S, C, out, argrow, fwin)
            switch (nch) {
                case 1: RF_PIPE_LAUNCH(1); break;
                case 2: RF_PIPE_LAUNCH(2); break;
                case 3: RF_PIPE_LAUNCH(3); break;
                default: RF_PIPE_LAUNCH(4); break;
            }
#undef RF_PIPE_LAUNCH
            return check_launch();
        }
    }
#define RF_FWD_LAUNCH(NCH)                                                                                        \
    if (!SURFACE && fwin)                                                                                          \
        hipLaunchKernelGGL((rf_fwd_kernel<SURFACE, NCH, !SURFACE, FT>), dim3(grid), dim3(RF_THREADS), lds, as_stream(stream), \
                           xyz, idx, dirs, fm, B, N, k, S, C, out, argrow, fwin);                                 \
    else                                                                                                           \
        hipLaunchKernelGGL((rf_fwd_kernel<SURFACE, NCH, false, FT>), dim3(grid), dim3(RF_THREADS), lds, as_stream(stream), \
                           xyz, idx, dirs, fm, B, N, k, S, C, out, argrow, fwin)
    switch (nch) {
        case 1: RF_FWD_LAUNCH(1); break;
        case 2: RF_FWD_LAUNCH(2); break;
        case 3: RF_FWD_LAUNCH(3); break;
        default: RF_FWD_LAUNCH(4); break;
    }
#undef RF_FWD_LAUNCH
    return check_launch();
}

extern "C" int hsp_rf_surface_fwd(const float* xyz, const int32_t* idx, const float* dirs_n, int B, int N, int k,
                                  int S, int K, float* out, uint16_t* argrow, hspStream_t stream) {
    return rf_fwd<true, float>(xyz, idx, dirs_n, nullptr, B, N, k, S, K, out, argrow, nullptr, stream);
}
extern "C" int hsp_rf_surface_fwd_bf16(const float* xyz, const int32_t* idx, const float* dirs_n, int B, int N, int k,
                                       int S, int K, hsp_bf16_t* out, uint16_t* argrow, hspStream_t stream) {
    return rf_fwd<true, bf16_t>(xyz, idx, dirs_n, nullptr, B, N, k, S, K, out, argrow, nullptr, stream);
}

// the forward's fwin stream pays off once a cloud's fm no longer stays in its XCD's L2 next to the other streams
// (round 3: always.  With the tile kernel's adds in fixed point the 4-byte gathers of the winners' support values were what
// the small clouds' backward waited on: 52.7 -> 28.1 us at N = 257, C = 256 for +5 us in the forward.)
extern "C" int hsp_rf_conv_wants_fwin(int N, int S, int C) {
    (void)N; (void)S; (void)C;
    return 1;
}

extern "C" int hsp_rf_conv_fwd(const float* xyz, const int32_t* idx, const float* dirs_n, const float* fm, int B,
                               int N, int k, int S, int C, float* out, uint16_t* argrow, float* fwin,
                               hspStream_t stream) {
    return rf_fwd<false, float>(xyz, idx, dirs_n, fm, B, N, k, S, C, out, argrow, fwin, stream);
}
extern "C" int hsp_rf_conv_fwd_bf16(const float* xyz, const int32_t* idx, const float* dirs_n, const hsp_bf16_t* fm, int B,
                                    int N, int k, int S, int C, hsp_bf16_t* out, uint16_t* argrow, hsp_bf16_t* fwin,
                                    hspStream_t stream) {
    return rf_fwd<false, bf16_t>(xyz, idx, dirs_n, fm, B, N, k, S, C, out, argrow, fwin, stream);
}
extern "C" int hsp_rf_conv_wants_fwin_bf16(int N, int S, int C) {
    return (size_t)N * (S + 1) * C * sizeof(bf16_t) >= ((size_t)3 << 20) ? 1 : 0;
}

extern "C" size_t hsp_rf_bwd_workspace_bytes(int SC) {
    if (SC <= 0) return 0;
    return (size_t)rf_bwd_grid_max(SC) * 3 * (size_t)SC * sizeof(float);
}

extern "C" int hsp_rf_conv_bwd(const float* xyz, const float* dirs_n, const float* fm, const uint16_t* argrow,
                               const float* grad_out, const int32_t* rev_off, const int32_t* rev_edge, int B, int N,
                               int k, int S, int C, float* grad_fm, float* grad_dirs_n, void* ws, size_t ws_bytes,
                               hspStream_t stream) {
    int rc = rf_check(xyz, dirs_n, fm, B, N, k, S, C);
    if (rc) return rc;
    if (!argrow || !grad_out || !rev_off || !rev_edge || !grad_fm || !grad_dirs_n) return HSP_ERR_BAD_ARG;
    const int SC = S * C;
    if (!ws || ws_bytes < hsp_rf_bwd_workspace_bytes(SC)) return HSP_ERR_WORKSPACE;
    const int nq = SC >> 2;
    const int nch = (nq + RF_THREADS - 1) / RF_THREADS;
    if (nch > 4) return HSP_ERR_UNSUPPORTED;              // S*C <= 4096
    hipStream_t st = as_stream(stream);
    const int grid = rf_bwd_grid((long long)B * N, SC);
    const size_t lds = (size_t)(RF_EB * (C + 8)) * 4;
    if (lds > 64 * 1024) return HSP_ERR_UNSUPPORTED;
    float* wsf = reinterpret_cast<float*>(ws);
#define RF_CSR_LAUNCH(NCH)                                                                                        \
    hipLaunchKernelGGL((rf_conv_bwd_csr_kernel<NCH>), dim3(grid), dim3(RF_THREADS), lds, st, xyz, dirs_n, fm, argrow, \
                       grad_out, rev_off, rev_edge, B, N, k, S, C, grad_fm, wsf)
    switch (nch) {
        case 1: RF_CSR_LAUNCH(1); break;
        case 2: RF_CSR_LAUNCH(2); break;
        case 3: RF_CSR_LAUNCH(3); break;
        default: RF_CSR_LAUNCH(4); break;
    }
#undef RF_CSR_LAUNCH
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(rf_dirs_reduce_kernel, dim3((SC + 63) / 64), dim3(1024), 0, st, wsf, grid, SC, dirs_n, grad_dirs_n);
    return check_launch();
}

// column-tile scatter: the widest tile whose acc[N][TC] + xyz[N][3] fits two workgroups per CU, else one
static int pick_tile_cols(int N, int C, bool surface, int B = 0, int SC = 0) {
    if (surface) return (C % 16 == 0) ? 16 : ((C % 8 == 0) ? 8 : 4);
    // small clouds take wider tiles (up to 64 columns): the per-workgroup set-up (direction normalisation, tile
    // zeroing / flush, gradient fold) is then amortised over 4x the elements and the row segments read are longer
    for (int pass = 0; pass < 2; ++pass)
        for (int tc = 64; tc >= 4; tc >>= 1) {
            if (tc > 16 && (long long)(SC / tc) * B < 2 * HSP_NUM_CU) continue;   // ... while the grid still fills the chip
            if (C % tc == 0 && ((size_t)N * tc + 3 * (size_t)N) * 4 <= (pass == 0 ? 80u : 156u) * 1024) return tc;
        }
    return 0;
}

extern "C" size_t hsp_rf_bwd_scatter_workspace_bytes(int B, int SC) {
    if (B <= 0 || SC <= 0) return 0;
    return (size_t)B * 2 * 3 * SC * sizeof(float);             // (x2: the two half-cloud tiles of the dense-cloud form)
}

// dense clouds: two half-cloud tiles of 16 columns instead of one whole-cloud tile of <= 8 (see rf_bwd_tile_kernel)
static bool rf_use_row_split(int N, int C, bool surface, int tc_whole) {
    if (surface || tc_whole >= 16 || C % 16) return false;
    const int nr = (N + 1) / 2;
    return ((size_t)nr * 16 + 3 * (size_t)nr) * 4 <= 156u * 1024;
}

template <bool SURFACE, bool FWIN, typename FT>
static int rf_bwd_scatter(const float* xyz, const float* dirs, const FT* fm, const uint16_t* argrow,
                          const FT* gout, int B, int N, int S, int C, FT* gfm, float* gdirs, void* ws,
                          size_t ws_bytes, hspStream_t stream, HspDirsPending* pending = nullptr) {
    int rc = rf_check(xyz, dirs, argrow, B, N, 1, S, C);
    if (rc) return rc;
    if (!gout || !gdirs || (!SURFACE && (!fm || !gfm))) return HSP_ERR_BAD_ARG;
    const int SC = S * C;
    if (!ws || ws_bytes < hsp_rf_bwd_scatter_workspace_bytes(B, SC)) return HSP_ERR_WORKSPACE;
    const int tc = pick_tile_cols(N, C, SURFACE, B, S * C);
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(ws);
    if (rf_use_row_split(N, C, SURFACE, tc)) {
        const int nr = (N + 1) / 2;
        const size_t lds2 = ((size_t)nr * 16 + 3 * (size_t)nr) * 4;
        auto kern = rf_bwd_tile_kernel<16, SURFACE, FWIN, FT, 2>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
        hipLaunchKernelGGL(kern, dim3(SC / 16, B, 2), dim3(RF_TILE_THREADS), lds2, st, xyz, dirs, fm, argrow, gout, B, N, S, C, gfm, part);
        rc = check_launch();
        if (rc) return rc;
        if (pending) { *pending = HspDirsPending{part, dirs, gdirs, 2 * B, SC}; return HSP_OK; }
        hipLaunchKernelGGL(rf_dirs_reduce_kernel, dim3((SC + 63) / 64), dim3(1024), 0, st, part, 2 * B, SC, dirs, gdirs);
        return check_launch();
    }
    if (!tc) return HSP_ERR_UNSUPPORTED;
    size_t lds = ((SURFACE ? 0 : (size_t)N * tc) + 3 * (size_t)N) * 4;
    if (lds < RF_TILE_THREADS * 12 * 4) lds = RF_TILE_THREADS * 12 * 4;
    dim3 grid(SC / tc, B);
#define RF_TILE_LAUNCH(TC)                                                                                          \
    {                                                                                                               \
        auto kern = rf_bwd_tile_kernel<TC, SURFACE, FWIN, FT>;                                                                \
        if (lds > 64 * 1024) {                                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }                                  \
        }                                                                                                           \
        hipLaunchKernelGGL(kern, grid, dim3(RF_TILE_THREADS), lds, st, xyz, dirs, fm, argrow, gout, B, N, S, C, gfm, part); \
    }
    if (tc == 64) RF_TILE_LAUNCH(64) else if (tc == 32) RF_TILE_LAUNCH(32) else if (tc == 16) RF_TILE_LAUNCH(16)
    else if (tc == 8) RF_TILE_LAUNCH(8) else RF_TILE_LAUNCH(4)
#undef RF_TILE_LAUNCH
    rc = check_launch();
    if (rc) return rc;
    if (pending) { *pending = HspDirsPending{part, dirs, gdirs, B, SC}; return HSP_OK; }
    hipLaunchKernelGGL(rf_dirs_reduce_kernel, dim3((SC + 63) / 64), dim3(1024), 0, st, part, B, SC, dirs, gdirs);
    return check_launch();
}

extern "C" int hsp_rf_surface_bwd(const float* xyz, const float* dirs_n, const uint16_t* argrow, const float* grad_out,
                                  int B, int N, int S, int K, float* grad_dirs_n, void* ws, size_t ws_bytes,
                                  hspStream_t stream) {
    return rf_bwd_scatter<true, false, float>(xyz, dirs_n, nullptr, argrow, grad_out, B, N, S, K, nullptr, grad_dirs_n, ws,
                                              ws_bytes, stream);
}
extern "C" int hsp_rf_surface_bwd_bf16(const float* xyz, const float* dirs_n, const uint16_t* argrow,
                                       const hsp_bf16_t* grad_out, int B, int N, int S, int K, float* grad_dirs_n, void* ws,
                                       size_t ws_bytes, hspStream_t stream) {
    return rf_bwd_scatter<true, false, bf16_t>(xyz, dirs_n, nullptr, argrow, grad_out, B, N, S, K, nullptr, grad_dirs_n, ws,
                                               ws_bytes, stream);
}

extern "C" int hsp_rf_conv_bwd_scatter(const float* xyz, const float* dirs_n, const float* fm, const float* fwin,
                                       const uint16_t* argrow, const float* grad_out, int B, int N, int S, int C,
                                       float* grad_fm, float* grad_dirs_n, void* ws, size_t ws_bytes,
                                       hspStream_t stream) {
    if (fwin)
        return rf_bwd_scatter<false, true, float>(xyz, dirs_n, fwin, argrow, grad_out, B, N, S, C, grad_fm, grad_dirs_n, ws,
                                                  ws_bytes, stream);
    return rf_bwd_scatter<false, false, float>(xyz, dirs_n, fm, argrow, grad_out, B, N, S, C, grad_fm, grad_dirs_n, ws,
                                               ws_bytes, stream);
}
extern "C" int hsp_rf_conv_bwd_scatter_bf16(const float* xyz, const float* dirs_n, const hsp_bf16_t* fm,
                                            const hsp_bf16_t* fwin, const uint16_t* argrow, const hsp_bf16_t* grad_out,
                                            int B, int N, int S, int C, hsp_bf16_t* grad_fm, float* grad_dirs_n, void* ws,
                                            size_t ws_bytes, hspStream_t stream) {
    if (fwin)
        return rf_bwd_scatter<false, true, bf16_t>(xyz, dirs_n, fwin, argrow, grad_out, B, N, S, C, grad_fm, grad_dirs_n, ws,
                                                   ws_bytes, stream);
    return rf_bwd_scatter<false, false, bf16_t>(xyz, dirs_n, fm, argrow, grad_out, B, N, S, C, grad_fm, grad_dirs_n, ws,
                                                ws_bytes, stream);
}

// the same backward launches with the direction-gradient fold left pending (hsp_step_fold runs it with the step's other folds)
extern "C" int hsp_rf_surface_bwd_partial(const float* xyz, const float* dirs_n, const uint16_t* argrow, const float* grad_out,
                                          int B, int N, int S, int K, float* grad_dirs_n, void* ws, size_t ws_bytes,
                                          HspDirsPending* pending, hspStream_t stream) {
    if (!pending) return HSP_ERR_BAD_ARG;
    return rf_bwd_scatter<true, false, float>(xyz, dirs_n, nullptr, argrow, grad_out, B, N, S, K, nullptr, grad_dirs_n, ws,
                                              ws_bytes, stream, pending);
}
extern "C" int hsp_rf_surface_bwd_partial_bf16(const float* xyz, const float* dirs_n, const uint16_t* argrow,
                                               const hsp_bf16_t* grad_out, int B, int N, int S, int K, float* grad_dirs_n,
                                               void* ws, size_t ws_bytes, HspDirsPending* pending, hspStream_t stream) {
    if (!pending) return HSP_ERR_BAD_ARG;
    return rf_bwd_scatter<true, false, bf16_t>(xyz, dirs_n, nullptr, argrow, grad_out, B, N, S, K, nullptr, grad_dirs_n, ws,
                                               ws_bytes, stream, pending);
}
extern "C" int hsp_rf_conv_bwd_scatter_partial(const float* xyz, const float* dirs_n, const float* fm, const float* fwin,
                                               const uint16_t* argrow, const float* grad_out, int B, int N, int S, int C,
                                               float* grad_fm, float* grad_dirs_n, void* ws, size_t ws_bytes,
                                               HspDirsPending* pending, hspStream_t stream) {
    if (!pending) return HSP_ERR_BAD_ARG;
    if (fwin)
        return rf_bwd_scatter<false, true, float>(xyz, dirs_n, fwin, argrow, grad_out, B, N, S, C, grad_fm, grad_dirs_n, ws,
                                                  ws_bytes, stream, pending);
    return rf_bwd_scatter<false, false, float>(xyz, dirs_n, fm, argrow, grad_out, B, N, S, C, grad_fm, grad_dirs_n, ws,
                                               ws_bytes, stream, pending);
}
extern "C" int hsp_rf_conv_bwd_scatter_partial_bf16(const float* xyz, const float* dirs_n, const hsp_bf16_t* fm,
                                                    const hsp_bf16_t* fwin, const uint16_t* argrow, const hsp_bf16_t* grad_out,
                                                    int B, int N, int S, int C, hsp_bf16_t* grad_fm, float* grad_dirs_n,
                                                    void* ws, size_t ws_bytes, HspDirsPending* pending, hspStream_t stream) {
    if (!pending) return HSP_ERR_BAD_ARG;
    if (fwin)
        return rf_bwd_scatter<false, true, bf16_t>(xyz, dirs_n, fwin, argrow, grad_out, B, N, S, C, grad_fm, grad_dirs_n, ws,
                                                   ws_bytes, stream, pending);
    return rf_bwd_scatter<false, false, bf16_t>(xyz, dirs_n, fm, argrow, grad_out, B, N, S, C, grad_fm, grad_dirs_n, ws,
                                                ws_bytes, stream, pending);
}
