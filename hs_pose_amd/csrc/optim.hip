// optim.hip -- fused multi-tensor optimizer step for gfx950.
//
// Replaces, for every parameter tensor at once:
//   torch.nn.utils.clip_grad_norm_(params, max_norm)                    engine/train.py:99,104
//   Ranger.step(): gradient centralisation -> RAdam -> Lookahead        tools/torch_utils/solver/ranger2020.py:135-246
// The reference walks 103 parameter tensors in Python and issues ~10 small elementwise kernels for each
// (~1000 launches per step, more than the whole forward + backward here).  All parameters, gradients and
// the three state tensors live in flat fp32 buffers; a table of ROWS (a dim-0 slice of a >= 2-D tensor,
// or a chunk of a 1-D tensor) drives one kernel in which a wave owns a row: it sums the row's gradient
// (the centralisation mean of ranger2020.py:31-41), then streams the update over the row.  The clip
// coefficient is read from device memory (hsp_sumsq_f32 -> no host round trip).
// HBM-bound: 28 B read + 16 B written per parameter (+4 B / 4 B on a Lookahead step).
#include "common.h"

namespace hsp {

struct RowDesc {            // mirrors HspRowDesc (include/hsp.h)
    long long offset;       // first element in the flat buffers
    int len;                // elements
    int gc;                 // subtract the row mean of the gradient first
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n,
                                                            float* __restrict__ part) {
    __shared__ float red[4];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const long long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = x4[i];
        a0 += v.x * v.x; a1 += v.y * v.y; a2 += v.z * v.z; a3 += v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) a0 += x[i] * x[i];
    float s = wave_sum((a0 + a1) + (a2 + a3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nblk,
                                                          float* __restrict__ out) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += part[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct RangerArgs {
    float beta1, beta2, eps, weight_decay, step_lr, la_alpha, max_norm;
    int adaptive, lookahead, gc_after;
};

// one wave per row
__global__ __launch_bounds__(256) void ranger_rows_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          float* __restrict__ slow,
                                                          const RowDesc* __restrict__ rows, int nrows,
                                                          const float* __restrict__ gnorm_sq, RangerArgs a) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const RowDesc rd = rows[r];
    float clip = 1.0f;
    if (gnorm_sq) {                                        // clip_grad_norm_: coef = max_norm / (norm + 1e-6), capped at 1
        const float c = a.max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f);
        clip = c < 1.0f ? c : 1.0f;
    }
    float* pp = p + rd.offset;
    const float* gp = g + rd.offset;
    float* mp = m + rd.offset;
    float* vp = v + rd.offset;
    float* sp = slow + rd.offset;
    // gradient centralisation (ranger2020.py:31-41, gc_loc=True): g -= mean over all dims but the first
    float mean = 0.f;
    if (rd.gc && !a.gc_after) {
        float s = 0.f;
        for (int e = lane; e < rd.len; e += 64) s += gp[e] * clip;
        mean = wave_sum(s) / (float)rd.len;
    }
    const float omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
    if (!(rd.gc && a.gc_after)) {
        for (int e = lane; e < rd.len; e += 64) {
            const float gr = gp[e] * clip - mean;
            const float vn = vp[e] * a.beta2 + omb2 * gr * gr;      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
            const float mn = mp[e] * a.beta1 + omb1 * gr;           // exp_avg.mul_(beta1).add_(g, alpha=1-beta1)
            vp[e] = vn;
            float G = a.adaptive ? mn / (sqrtf(vn) + a.eps) : mn;   // ranger2020.py:215-219
            float w = pp[e];
            if (a.weight_decay != 0.f) G += a.weight_decay * w;
            // on the non-adaptive steps the reference's G_grad IS exp_avg (:219), so its in-place weight-decay
            // term also lands in the stored moment: reproduced
            mp[e] = a.adaptive ? mn : G;
            w -= a.step_lr * G;                                     // p.add_(G, alpha=-step_size*lr)
            if (a.lookahead) {                                      // ranger2020.py:232-238
                const float sl = sp[e] + a.la_alpha * (w - sp[e]);
                sp[e] = sl;
                w = sl;
            }
            pp[e] = w;
        }
        return;
    }
    // gc_loc=False: the centralisation acts on the generalised gradient G (ranger2020.py:223-225): two sweeps.
    // (non-adaptive steps: G aliases exp_avg in the reference, so the stored moment ends up centralised too)
    float s = 0.f;
    for (int e = lane; e < rd.len; e += 64) {
        const float gr = gp[e] * clip;
        const float vn = vp[e] * a.beta2 + omb2 * gr * gr;
        const float mn = mp[e] * a.beta1 + omb1 * gr;
        vp[e] = vn;
        float G = a.adaptive ? mn / (sqrtf(vn) + a.eps) : mn;
        if (a.weight_decay != 0.f) G += a.weight_decay * pp[e];
        mp[e] = a.adaptive ? mn : G;
        s += G;
    }
    const float gmean = wave_sum(s) / (float)rd.len;
    for (int e = lane; e < rd.len; e += 64) {
        float w = pp[e];
        float G;
        if (a.adaptive) {
            G = mp[e] / (sqrtf(vp[e]) + a.eps);
            if (a.weight_decay != 0.f) G += a.weight_decay * w;
            G -= gmean;
        } else {
            G = mp[e] - gmean;
            mp[e] = G;
        }
        w -= a.step_lr * G;
        if (a.lookahead) {
            const float sl = sp[e] + a.la_alpha * (w - sp[e]);
            sp[e] = sl;
            w = sl;
        }
        pp[e] = w;
    }
}

static int sumsq_blocks(long long n) {
    long long b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}

}  // namespace hsp

using namespace hsp;

extern "C" size_t hsp_sumsq_workspace_bytes(long long n) {
    if (n <= 0) return 0;
    return (size_t)sumsq_blocks(n) * sizeof(float);
}

extern "C" int hsp_sumsq_f32(const float* x, long long n, float* out, void* ws, size_t ws_bytes, hspStream_t stream) {
    if (!x || !out || n <= 0) return HSP_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return HSP_ERR_UNSUPPORTED;     // float4 sweeps
    if (!ws || ws_bytes < hsp_sumsq_workspace_bytes(n)) return HSP_ERR_WORKSPACE;
    const int nb = sumsq_blocks(n);
    float* part = reinterpret_cast<float*>(ws);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, x, n, part);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, part, nb, out);
    return check_launch();
}

extern "C" int hsp_ranger_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* slow,
                               const HspRowDesc* rows, int nrows, float beta1, float beta2, float eps,
                               float weight_decay, float step_lr, int adaptive, int lookahead, float la_alpha,
                               int gc_after, const float* gnorm_sq, float max_norm, hspStream_t stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !slow || !rows || nrows <= 0) return HSP_ERR_BAD_ARG;
    static_assert(sizeof(RowDesc) == sizeof(HspRowDesc), "row descriptor layout");
    RangerArgs a;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.step_lr = step_lr;
    a.la_alpha = la_alpha; a.max_norm = max_norm; a.adaptive = adaptive; a.lookahead = lookahead; a.gc_after = gc_after;
    hipLaunchKernelGGL(ranger_rows_kernel, dim3((nrows + 3) / 4), dim3(256), 0, as_stream(stream), params, grads, exp_avg,
                       exp_avg_sq, slow, reinterpret_cast<const RowDesc*>(rows), nrows, gnorm_sq, a);
    return check_launch();
}
