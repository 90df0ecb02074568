// exact.hip -- the reductions and element-wise formulas of the reference's forward whose ROUNDING ORDER decides which
// neighbours the next layer's feature-space KNN picks (gcn3d.py:19-23 ranks rows that differ in the last bits), restated in the
// exact order the reference's CPU path (ATen, torch 2.x) evaluates them.  Together with the k-ordered products of gemm_wave.hip
// they make the eval-mode forward of the HS stack reproduce the reference's feature rows bit for bit, so a free-running forward
// selects the reference's neighbour lists instead of lists that agree "up to near ties" (DESIGN.md section 2.2).
// Pinned by tests/golden/exact_*.npz (oracle/gen_golden_exact.py imports the reference and records its outputs).
#include "common.h"

extern "C" int hsp_gather_max_fwd(const float* feat, const int32_t* idx, const int32_t* qsel, int B, int Nsrc, int Nidx, int Nq, int k,
                                  int kstride, int C, float* out, uint8_t* argmax, hspStream_t stream);

namespace hsp {

// ---- get_ORL_global (gcn3d.py:211-218): torch.mean(max_n feature[idx], dim=1) ------------------------------------------------
// ATen sums an outer (strided) dimension with cascade_sum / multi_row_sum: rows are added one by one into a level-0 accumulator
// that is dumped into level 1 every 16 rows, level 1 into level 2 every 256, level 2 into level 3 every 4096 (level_step = 16 for
// every N < 2^20), the remainder rows go into level 0 last, and the levels are added 0 <- 1 <- 2 <- 3; the mean divides by N.
// Level 0 is 16-row chunks summed from zero: one thread per (chunk, 4 channels).
#define ORLX_ROWS 16
// (the neighbourhood max G (B,N,C) comes from hsp_gather_max_fwd -- a thread per (point, 4 channels) --, then:)
__global__ __launch_bounds__(256) void orl_exact_l0_kernel(const float* __restrict__ G, int N, int C, float* __restrict__ part,
                                                           int nchunk) {
    const int cq = C >> 2;
    const int b = blockIdx.y;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)nchunk * cq) return;
    const int chunk = (int)(e / cq), g = (int)(e - (long long)chunk * cq);
    const int r0 = chunk * ORLX_ROWS, r1 = min(N, r0 + ORLX_ROWS);
    const float* gb = G + (size_t)b * N * C + (g << 2);
    float4 v[ORLX_ROWS];
#pragma unroll
    for (int t = 0; t < ORLX_ROWS; ++t) v[t] = *reinterpret_cast<const float4*>(gb + (size_t)min(r0 + t, N - 1) * C);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < ORLX_ROWS; ++t)
        if (r0 + t < r1) { s.x = add_rn(s.x, v[t].x); s.y = add_rn(s.y, v[t].y); s.z = add_rn(s.z, v[t].z); s.w = add_rn(s.w, v[t].w); }
    *reinterpret_cast<float4*>(part + ((size_t)b * nchunk + chunk) * C + (g << 2)) = s;
}

// levels 1..3 and the division: one thread per (cloud, channel)
__global__ __launch_bounds__(256) void orl_exact_fold_kernel(const float* __restrict__ part, int B, int N, int nchunk, int C,
                                                             float* __restrict__ fg) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * C) return;
    const int b = e / C, c = e - b * C;
    const float* p = part + (size_t)b * nchunk * C + c;
    const int nfull = N / ORLX_ROWS;
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int j = 0; j < nfull; ++j) {
        const int i = (j + 1) * ORLX_ROWS;                 // rows consumed so far
        a1 = add_rn(a1, p[(size_t)j * C]);
        if ((i & 0xf0) == 0) {
            a2 = add_rn(a2, a1); a1 = 0.f;
            if ((i & 0xf00) == 0) { a3 = add_rn(a3, a2); a2 = 0.f; }
        }
    }
    float s = nfull < nchunk ? p[(size_t)nfull * C] : 0.f; // the remainder rows (level 0 after the last dump)
    s = add_rn(s, a1); s = add_rn(s, a2); s = add_rn(s, a3);
    fg[e] = __fdiv_rn(s, (float)N);
}

// ---- eval-mode BatchNorm1d on the reference's transposed (B,C,N) view (FaceRecon.py:90-95) ------------------------------------
// ATen's generic path (batch_norm_cpu_transform_input_template): invstd = 1 / at::sqrt(running_var + eps), then
// ((x - mean) * invstd) * weight + bias, each operation rounded (no fma).  at::sqrt on the CPU is MKL VML's vsSqrt: within an ulp
// but NOT correctly rounded (one channel in ~180 differs from the IEEE value: oracle/gen_golden_exact.py), so the caller may
// hand in the invstd the host's ATen computed (``invstd`` != NULL); without it the correctly rounded value is used.
__global__ __launch_bounds__(256) void bn_eval_exact_kernel(const float* __restrict__ x, long long n4, int C,
                                                            const float* __restrict__ rm, const float* __restrict__ rv,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ w, const float* __restrict__ bia, float eps,
                                                            int relu, float* __restrict__ y) {
    const int cq = C >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % cq) << 2;
        const float4 v = *reinterpret_cast<const float4*>(x + e * 4);
        const float4 m = *reinterpret_cast<const float4*>(rm + c);
        float4 var = make_float4(1.f, 1.f, 1.f, 1.f), iv = var;
        if (invstd) iv = *reinterpret_cast<const float4*>(invstd + c);
        else var = *reinterpret_cast<const float4*>(rv + c);
        const float4 ww = *reinterpret_cast<const float4*>(w + c), bb = *reinterpret_cast<const float4*>(bia + c);
        float4 o;
#define BNX(X)                                                                       \
        {                                                                            \
            const float inv = invstd ? iv.X : __fdiv_rn(1.0f, sqrtf(add_rn(var.X, eps)));   \
            o.X = add_rn(mul_rn(mul_rn(sub_rn(v.X, m.X), inv), ww.X), bb.X);         \
            if (relu) o.X = fmaxf(o.X, 0.f);                                         \
        }
        BNX(x) BNX(y) BNX(z) BNX(w)
#undef BNX
        *reinterpret_cast<float4*>(y + e * 4) = o;
    }
}

// ---- ATen's sums over a STRIDED dimension (sum_kernel_impl -> vectorized_outer_sum) ----------------------------------------------
// Reducing a dimension of stride != 1 while another dimension is contiguous walks the contiguous one in columns: the first
// 32 * floor(cols / 32) columns (4 vectors of 8 lanes) are summed by multi_row_sum = the plain cascade (level-0 accumulator
// dumped every 16 elements, level 1 every 256, level 2 every 4096, levels added 0 <- 1 <- 2 <- 3); the REMAINING columns by
// row_sum: four interleaved partial sums (elements e = 4 i + p, each partial a cascade over floor(size / 4) elements), the
// size % 4 leftover elements into partial 0, then partial 0 += 1, += 2, += 3.
__device__ __forceinline__ int ceil_log2_int(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

template <typename F>   // F(e) -> element e of the reduced dimension
__device__ float aten_cascade(F elem, int first, int step, int size) {
    const int lp = max(4, ceil_log2_int(size) / 4), ls = 1 << lp, lm = ls - 1;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    for (; i + ls <= size;) {
        for (int j = 0; j < ls; ++j, ++i) a0 = add_rn(a0, elem(first + i * step));
        a1 = add_rn(a1, a0); a0 = 0.f;
        if ((i & (lm << lp)) == 0) {
            a2 = add_rn(a2, a1); a1 = 0.f;
            if ((i & (lm << (2 * lp))) == 0) { a3 = add_rn(a3, a2); a2 = 0.f; }
        }
    }
    for (; i < size; ++i) a0 = add_rn(a0, elem(first + i * step));
    a0 = add_rn(a0, a1); a0 = add_rn(a0, a2); a0 = add_rn(a0, a3);
    return a0;
}
template <typename F>
__device__ float aten_row_sum_ilp4(F elem, int size) {
    const int si = size / 4;
    float p[4];
    for (int kk = 0; kk < 4; ++kk) p[kk] = aten_cascade(elem, kk, 4, si);
    for (int i = si * 4; i < size; ++i) p[0] = add_rn(p[0], elem(i));
    p[0] = add_rn(p[0], p[1]); p[0] = add_rn(p[0], p[2]); p[0] = add_rn(p[0], p[3]);
    return p[0];
}
template <typename F>
__device__ float aten_outer_sum(F elem, int size, int col, int cols) {
    return col < (cols / 32) * 32 ? aten_cascade(elem, 0, 1, size) : aten_row_sum_ilp4(elem, size);
}

// |x|^2 per row of x (B,N,C) as torch.sum(v ** 2, dim=2) evaluates it when v is the TRANSPOSED VIEW of a (B,C,N) tensor --
// what FaceRecon.py:94-95 hands conv_3: relu(bn(...)).transpose(1, 2) is never made contiguous, so the channel sum is an
// outer sum over columns n (gcn3d.py:20)
// One wave per 64 rows: the rows are staged whole in LDS with coalesced loads (a thread walking its own row read one float per 1 KB
// stride, 256 dependent misses: 37 us for 1028 x 256 -- the longest small kernel of an inference forward), then lane r runs row r's
// sum in ATen's order from LDS (pitch C + 1: the lanes' walks fall on different banks).
__global__ __launch_bounds__(64) void quad_outer_kernel(const float* __restrict__ x, int N, int C, float* __restrict__ quad) {
    extern __shared__ float qo_tile[];                 // 64 x (C + 1)
    const int lane = threadIdx.x, b = blockIdx.y, n0 = blockIdx.x * 64;
    const int P = C + 1;
    const float* xb = x + ((size_t)b * N + n0) * C;
    const int nrows = min(64, N - n0);
    for (int e = lane; e < nrows * C; e += 64) {       // consecutive lanes, consecutive floats
        const int r = e / C, c = e - r * C;
        qo_tile[r * P + c] = xb[e];
    }
    __builtin_amdgcn_wave_barrier();
    const int n = n0 + lane;
    if (lane >= nrows) return;
    const float* row = qo_tile + lane * P;
    quad[(size_t)b * N + n] = aten_outer_sum([&](int c) { const float v = row[c]; return mul_rn(v, v); }, C, n, N);
}

// (rows too wide for the LDS tile: a thread per row, straight from global memory)
__global__ __launch_bounds__(256) void quad_outer_wide_kernel(const float* __restrict__ x, int N, int C, float* __restrict__ quad) {
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= N) return;
    const float* row = x + ((size_t)b * N + n) * C;
    quad[(size_t)b * N + n] = aten_outer_sum([&](int c) { const float v = __ldg(row + c); return mul_rn(v, v); }, C, n, N);
}

// PoseNet9D.py:25: points - points.mean(dim=1, keepdim=True): the mean over the N points of a contiguous (B,N,3) tensor is an
// outer sum over 3 columns (all of them "remaining" columns: the interleaved form), divided by N
__global__ __launch_bounds__(256) void center_cloud_kernel(const float* __restrict__ pts, int N, float* __restrict__ out,
                                                           float* __restrict__ mean) {
    // column c (0..2), partial p (0..3): elements e = 4 i + p, i < si = N / 4, summed as a cascade -> 12 cascades whose 16-element
    // level-0 chunks are independent: one thread per (cascade, chunk), then one thread per cascade folds the levels
    extern __shared__ float sc[];                      // 12 x nchunk chunk sums, then 12 cascade results, then 3 means
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = pts + (size_t)b * N * 3;
    const int si = N / 4;
    const int lp = max(4, ceil_log2_int(si) / 4), ls = 1 << lp;
    const int nfull = si / ls, nchunk = nfull + 1;     // (the last chunk: the si % ls remainder elements, maybe none)
    for (int w = tid; w < 12 * nchunk; w += 256) {
        const int cas = w / nchunk, ch = w - cas * nchunk, c = cas >> 2, pp = cas & 3;
        const int i0 = ch * ls, i1 = min(si, i0 + ls);
        float a = 0.f;
        for (int i = i0; i < i1; ++i) a = add_rn(a, p[(size_t)(4 * i + pp) * 3 + c]);
        sc[w] = a;
    }
    __syncthreads();
    float* res = sc + 12 * nchunk;
    if (tid < 12) {
        const float* cs = sc + tid * nchunk;
        const int lm = ls - 1;
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int j = 0; j < nfull; ++j) {
            const int i = (j + 1) * ls;
            a1 = add_rn(a1, cs[j]);
            if ((i & (lm << lp)) == 0) {
                a2 = add_rn(a2, a1); a1 = 0.f;
                if ((i & (lm << (2 * lp))) == 0) { a3 = add_rn(a3, a2); a2 = 0.f; }
            }
        }
        float a0 = cs[nfull];
        a0 = add_rn(a0, a1); a0 = add_rn(a0, a2); a0 = add_rn(a0, a3);
        res[tid] = a0;
    }
    __syncthreads();
    if (tid < 3) {
        float s0 = res[tid * 4];
        for (int i = si * 4; i < N; ++i) s0 = add_rn(s0, p[(size_t)i * 3 + tid]);     // the N % 4 leftover elements into partial 0
        s0 = add_rn(s0, res[tid * 4 + 1]); s0 = add_rn(s0, res[tid * 4 + 2]); s0 = add_rn(s0, res[tid * 4 + 3]);
        const float m = __fdiv_rn(s0, (float)N);
        res[12 + tid] = m;
        mean[b * 3 + tid] = m;
    }
    __syncthreads();
    for (int e = tid; e < N * 3; e += 256) out[(size_t)b * N * 3 + e] = sub_rn(p[e], res[12 + e % 3]);
}

}  // namespace hsp

using namespace hsp;

extern "C" size_t hsp_orl_exact_workspace_bytes(int B, int N, int C) {
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    return (size_t)B * ((N + ORLX_ROWS - 1) / ORLX_ROWS) * C * sizeof(float) + (size_t)B * N * C * sizeof(float);   // chunk sums + G
}

extern "C" int hsp_orl_global_exact_f32(const float* feat, const int32_t* idx, int B, int N, int k, int kstride, int C, float* fg,
                                        uint8_t* argmax, void* ws, size_t ws_bytes, hspStream_t stream) {
    if (!feat || !idx || !fg || !argmax || B <= 0 || N <= 0 || k <= 0 || kstride < k || C <= 0) return HSP_ERR_BAD_ARG;
    if ((C & 3) || k > 255 || N >= (1 << 20)) return HSP_ERR_UNSUPPORTED;        // (level_step = 16 holds below 2^20 rows)
    if (!ws || ws_bytes < hsp_orl_exact_workspace_bytes(B, N, C)) return HSP_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    const int nchunk = (N + ORLX_ROWS - 1) / ORLX_ROWS;
    float* part = reinterpret_cast<float*>(ws);
    float* G = part + (size_t)B * nchunk * C;
    const int rc = hsp_gather_max_fwd(feat, idx, nullptr, B, N, N, N, k, kstride, C, G, argmax, stream);
    if (rc) return rc;
    const long long work = (long long)nchunk * (C >> 2);
    hipLaunchKernelGGL(orl_exact_l0_kernel, dim3((unsigned)((work + 255) / 256), B), dim3(256), 0, st, G, N, C, part, nchunk);
    hipLaunchKernelGGL(orl_exact_fold_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, part, B, N, nchunk, C, fg);
    return check_launch();
}

extern "C" int hsp_bn_eval_f32(const float* x, long long R, int C, const float* running_mean, const float* running_var,
                               const float* invstd, const float* weight, const float* bias, float eps, int relu, float* y,
                               hspStream_t stream) {
    if (!x || !y || !running_mean || (!running_var && !invstd) || !weight || !bias || R <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    if (C & 3) return HSP_ERR_UNSUPPORTED;
    const long long n4 = R * (C >> 2);
    long long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_eval_exact_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, n4, C, running_mean,
                       running_var, invstd, weight, bias, eps, relu, y);
    return check_launch();
}

extern "C" int hsp_quad_outer_f32(const float* x, int B, int N, int C, float* quad, hspStream_t stream) {
    if (!x || !quad || B <= 0 || N <= 0 || C <= 0) return HSP_ERR_BAD_ARG;
    const size_t lds = (size_t)64 * (C + 1) * sizeof(float);
    if (lds <= 64 * 1024 + 1024 && lds <= 160 * 1024) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(quad_outer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
        }
        hipLaunchKernelGGL(quad_outer_kernel, dim3((N + 63) / 64, B), dim3(64), lds, as_stream(stream), x, N, C, quad);
    } else {
        hipLaunchKernelGGL(quad_outer_wide_kernel, dim3((N + 255) / 256, B), dim3(256), 0, as_stream(stream), x, N, C, quad);
    }
    return check_launch();
}

extern "C" int hsp_center_cloud_f32(const float* pts, int B, int N, float* centred, float* mean, hspStream_t stream) {
    if (!pts || !centred || !mean || B <= 0 || N <= 0) return HSP_ERR_BAD_ARG;
    int lp = 4, si = N / 4, l2 = 0;
    while ((1 << l2) < si) ++l2;
    if (l2 / 4 > lp) lp = l2 / 4;
    const size_t lds = (size_t)(12 * (si / (1 << lp) + 1) + 16) * sizeof(float);
    hipLaunchKernelGGL(center_cloud_kernel, dim3(B), dim3(256), lds, as_stream(stream), pts, N, centred, mean);
    return check_launch();
}
