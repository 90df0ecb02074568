// rfconv_mfma.hip -- receptive-field graph convolution forward with theta = R . D^ on the fp32 matrix cores (gfx950).
//
// Same contract and the same bits as rf_fwd_pipe_kernel (rfconv.hip): HSlayer_surface.graph_conv / HS_layer.graph_conv of the
// reference (network/fs_net_repo/gcn3d.py:92-107, :158-181).  What changes is who does the arithmetic and in which layout:
//
//   * theta[n, j] = R[n,:] . D^[:,j] for the k neighbours of a point and 32 support columns is ONE 32x32 tile of
//     v_mfma_f32_32x32x2_f32 (K = 3 padded to 4: two instructions).  That instruction IS a k-ordered fp32 fma chain
//     (fma(Rz,Dz, fma(Ry,Dy, fma(Rx,Dx,0))); the padded fourth term adds 0*finite), i.e. bit for bit the chain the VALU form
//     evaluates and the reference's matmul (gcn3d.py:101,167).  12 of the 37 VALU instructions per (neighbour, float4 of
//     columns) of the VALU form leave the VALU (the kernel was VALU-issue-bound: DESIGN.md section 8).
//   * the 32x32 C/D layout puts ONE column and 16 of the 32 rows in a lane (lanes 0-31: rows {0-3, 8-11, 16-19, 24-27}, lanes
//     32-63 the others).  The neighbours are dealt to the rows so that each half-wave holds ceil(k/2) of them in its FIRST
//     registers: the max over the neighbours is then an in-lane v_max3 chain (exact), the arg-max an equality scan in reverse
//     order (first winner wins, like the strict '>' of the VALU form), and the two halves meet through one
//     v_permlane32_swap.  The winner's support value is fetched after the scan (one 4-byte gather per lane and tile instead of
//     a select per neighbour), its row id comes from LDS.
//   * a wave owns (point, 32 channels) and walks the S supports of those channels: the mean over the supports is an in-lane
//     sum in support order -- no LDS round trip, no second barrier.
//   * the C channels are walked in passes of C / nsplit channels (all S supports of a channel stay in one pass): the gather
//     working set of a pass, N * S * C / nsplit elements, is chosen to fit the XCD's 4 MiB L2 next to the write-once streams
//     (N = 1028, C = 128 fp32: 3.7 MB in one pass -- measured 4.4x re-fetch -- 1.8 MB in two).  Clouds stay pinned to XCDs.
//   * a fifth wave per workgroup builds the unit directions / row offsets of the NEXT iteration's points into the other half
//     of a double buffer (two dependent global round trips + the sqrt / division chain of F.normalize) while the four
//     consumer waves work: one barrier per iteration.
//
// Workgroup = 320 threads; an iteration = 4 (point, 32-channel chunk) items = P = 4 / CPP points, CPP = C / (32 nsplit) in {1,2,4}.
// dynamic LDS: 3 S C floats (normalised directions) + 64 (zeros) + 2 buffers x P x (128 floats + 32 x 8 bytes).
#include "common.h"

namespace hsp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define RFM_THREADS 256

#ifdef RF_RELU_MAX
#define RFM_RELU(X) fmaxf((X), 0.f)
#else
#define RFM_RELU(X) __builtin_amdgcn_fmed3f((X), 0.f, 1.f)      // == rfconv.hip RF_RELU
#endif

// MFMA C/D row of accumulator register r in half-wave h (guide section 3): (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ int rfm_row(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <typename FT> __device__ __forceinline__ float rfm_gather(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ float rfm_gather<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
}
template <> __device__ __forceinline__ float rfm_gather<bf16_t>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)voff, (int)soff, 0) << 16);
}
template <typename FT> __device__ __forceinline__ void rfm_store_nt(__amdgpu_buffer_rsrc_t rs, float v, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ void rfm_store_nt<float>(__amdgpu_buffer_rsrc_t rs, float v, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)voff, (int)soff, 2);
}
template <> __device__ __forceinline__ void rfm_store_nt<bf16_t>(__amdgpu_buffer_rsrc_t rs, float v, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)f32_to_bf16_bits(v), rs, (int)voff, (int)soff, 2);
}

// a wave-uniform pointer the compiler cannot prove uniform (derived from the wave index): through v_readfirstlane, so that
// buffer descriptors built from it live in SGPRs without a waterfall loop (guide T20)
template <typename T> __device__ __forceinline__ T* rfm_uniform(const T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    return reinterpret_cast<T*>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v));
}

struct RfmItem { int ci, pass, g; };

// NC feature values of one lane (its NC adjacent columns) from / to a buffer: ONE access of NC * sizeof(FT) bytes
template <typename FT, int NC> struct RfmVec;
template <> struct RfmVec<float, 2> {
    typedef u32x2 raw_t;
    static __device__ __forceinline__ raw_t gather(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
        return __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0);
    }
    static __device__ __forceinline__ float get(const raw_t& v, int q) { return __uint_as_float(v[q]); }
    static __device__ __forceinline__ void store_nt(__amdgpu_buffer_rsrc_t rs, const float (&v)[2], unsigned voff, unsigned soff) {
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}, rs, (int)voff, (int)soff, 2);
    }
};
template <> struct RfmVec<bf16_t, 2> {
    typedef unsigned raw_t;
    static __device__ __forceinline__ raw_t gather(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
        return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0);
    }
    static __device__ __forceinline__ float get(const raw_t& v, int q) { return __uint_as_float(q ? (v & 0xffff0000u) : (v << 16)); }
    static __device__ __forceinline__ void store_nt(__amdgpu_buffer_rsrc_t rs, const float (&v)[2], unsigned voff, unsigned soff) {
        __builtin_amdgcn_raw_buffer_store_b32(f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16), rs, (int)voff, (int)soff, 2);
    }
};
template <> struct RfmVec<bf16_t, 4> {
    typedef u32x2 raw_t;
    static __device__ __forceinline__ raw_t gather(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
        return __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0);
    }
    static __device__ __forceinline__ float get(const raw_t& v, int q) {
        const unsigned w = v[q >> 1];
        return __uint_as_float((q & 1) ? (w & 0xffff0000u) : (w << 16));
    }
    static __device__ __forceinline__ void store_nt(__amdgpu_buffer_rsrc_t rs, const float (&v)[4], unsigned voff, unsigned soff) {
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16),
                                                    f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16)}, rs, (int)voff, (int)soff, 2);
    }
};
template <typename FT> __device__ __forceinline__ void rfm_store1_nt(__amdgpu_buffer_rsrc_t rs, float v, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ void rfm_store1_nt<float>(__amdgpu_buffer_rsrc_t rs, float v, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)voff, (int)soff, 2);
}
template <> __device__ __forceinline__ void rfm_store1_nt<bf16_t>(__amdgpu_buffer_rsrc_t rs, float v, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)f32_to_bf16_bits(v), rs, (int)voff, (int)soff, 2);
}

// One WAVE owns a (point, 32 NC channels) item at a time and nothing is shared between the waves of a workgroup but the
// normalised directions: no barrier after the prologue.  NC: adjacent columns per lane (NC theta tiles per support share one
// gather per neighbour: with one column per lane the texture addresser, not the VALU, set the pace -- measured TA 79 % busy).
// ST > 0: the number of supports at compile time (the tile loop unrolls: exact s_waitcnt counts); ST == 0: S at run time.
// WF: also write the winners' support values (fwin).
// dynamic LDS: 3 S C floats (directions) + per wave 32 x 8 bytes (row table) + [WF] 64 KH NC floats (the gathered values, for
// the winner's look-up)
template <bool SURFACE, bool WF, typename FT, int NC, int KH, int ST>
__global__ __launch_bounds__(RFM_THREADS) void rf_fwd_mfma_kernel(const float* __restrict__ xyz,
                                                                  const int32_t* __restrict__ idx,
                                                                  const float* __restrict__ dirs,
                                                                  const FT* __restrict__ fm, int B, int N, int k, int S_rt,
                                                                  int C, int nsplit, FT* __restrict__ out,
                                                                  uint16_t* __restrict__ argrow, FT* __restrict__ fwin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr unsigned ES = (unsigned)sizeof(FT);
    constexpr int W = 32 * NC;                                    // columns of an item
    constexpr int NWV = RFM_THREADS / 64;
    typedef RfmVec<FT, NC> Vec;
    typedef typename Vec::raw_t raw_t;
    const int S = ST > 0 ? ST : S_rt;
    const int SC = S * C;
    const int CH = C / nsplit, CPP = CH / W;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    float* sDx = reinterpret_cast<float*>(smem);
    float* sDy = sDx + SC;
    float* sDz = sDy + SC;
    u32x2* sRow = reinterpret_cast<u32x2*>(sDz + SC) + wave * 32;     // this wave's {row byte offset in the cloud's fm, row id} x 32
    float* sF = reinterpret_cast<float*>(reinterpret_cast<u32x2*>(sDz + SC) + NWV * 32) + wave * (64 * KH * NC);
    const int fstride = (S + 1) * C;

    // ---- this wave's share of the (cloud, pass, point, chunk) sequence of its XCD ------------------------------------------
    int b0, bstride, nb, vw, nvw;
    {
        const int xcd = blockIdx.x % HSP_NUM_XCD, local = blockIdx.x / HSP_NUM_XCD, per_xcd = gridDim.x / HSP_NUM_XCD;
        if (B >= HSP_NUM_XCD) {
            b0 = xcd; bstride = HSP_NUM_XCD; nb = (B - xcd + HSP_NUM_XCD - 1) / HSP_NUM_XCD; vw = local; nvw = per_xcd;
        } else {                                   // fewer clouds than XCDs: XCDs x, x + B, ... share cloud x % B
            b0 = xcd % B; bstride = 0; nb = 1;
            const int share = xcd / B, nshare = (HSP_NUM_XCD - b0 + B - 1) / B;
            vw = local + per_xcd * share; nvw = per_xcd * nshare;
        }
        vw = vw * NWV + wave; nvw *= NWV;
    }

    // ---- once per workgroup: F.normalize(directions, dim=0) (gcn3d.py:100,166) into LDS -------------------------------------
    for (int j = tid; j < SC; j += RFM_THREADS) {
        const float x = dirs[j], y = dirs[SC + j], z = dirs[2 * SC + j];
        const float n2 = add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z));
        const float nr = fmaxf(__fsqrt_rn(n2), 1e-12f);
        sDx[j] = __fdiv_rn(x, nr); sDy[j] = __fdiv_rn(y, nr); sDz[j] = __fdiv_rn(z, nr);
    }
    __syncthreads();

    // position u of the sequence as mixed-radix digits (cloud ci, pass, point i, chunk qc), advanced by nvw without divisions
    struct Pos { int ci, pass, i, qc; };
    auto decode = [&](int u) {
        Pos p;
        p.qc = u % CPP; u /= CPP;
        p.i = u % N; u /= N;
        p.pass = u % nsplit; p.ci = u / nsplit;
        return p;
    };
    const Pos dlt = decode(nvw);
    auto advance = [&](Pos p) {
        p.qc += dlt.qc; if (p.qc >= CPP) { p.qc -= CPP; ++p.i; }
        p.i += dlt.i; if (p.i >= N) { p.i -= N; ++p.pass; }
        p.pass += dlt.pass; if (p.pass >= nsplit) { p.pass -= nsplit; ++p.ci; }
        p.ci += dlt.ci;
        return p;
    };

    // ---- the point's neighbour of this lane's MFMA row: slot (half hh, register rr) holds neighbour hh * KH + rr; a slot past
    // the list (k < 2 KH, or one of the 16 - KH spare registers) repeats the first neighbour of its half (of the list, when the
    // half is empty): a repeat ties with its original, never beats it, and sits behind it in the scan order and in the half
    // order -- so nothing needs masking.
    const int cnt0 = k < KH ? k : KH, cnt1 = k - cnt0;
    const int hh_row = (l31 >> 2) & 1, rr_row = (l31 & 3) | ((l31 >> 3) << 2);      // inverse of rfm_row
    const int n_row = hh_row ? (rr_row < cnt1 ? KH + rr_row : (cnt1 > 0 ? KH : 0)) : (rr_row < cnt0 ? rr_row : 0);
    auto load_m = [&](const Pos& p) {                                 // neighbour id of this lane's row (0 past the sequence)
        int m = 0;
        if (p.ci < nb) m = idx[((size_t)(b0 + bstride * p.ci) * N + p.i) * k + n_row];
        return m;
    };
    auto load_q = [&](const Pos& p, int m, float (&q)[3]) {
        const float* xb = xyz + (size_t)(b0 + bstride * (p.ci < nb ? p.ci : 0)) * N * 3;
        q[0] = xb[m * 3]; q[1] = xb[m * 3 + 1]; q[2] = xb[m * 3 + 2];
    };
    // unit direction of the row's neighbour (gcn3d.py:49-59) -> this lane's A operands (Rx | Ry), (Rz | 0); row table to LDS
    auto finish = [&](const Pos& p, int m, const float (&q)[3], float& a1, float& a2) {
        const float* xp = xyz + ((size_t)(b0 + bstride * (p.ci < nb ? p.ci : 0)) * N + p.i) * 3;
        const float3 r = unit_dir(xp[0], xp[1], xp[2], q[0], q[1], q[2]);
        a1 = h ? r.y : r.x;
        a2 = h ? 0.f : r.z;
        if (h == 0 && rr_row < KH) sRow[hh_row * 16 + rr_row] = u32x2{(unsigned)m * (unsigned)fstride * ES, (unsigned)m};
    };

    const int total = nb * nsplit * N * CPP;
    if (vw >= total) return;
    Pos pc = decode(vw);                    // current item
    Pos pn = advance(pc);                   // next
    float a1, a2;
    int m_n;                                // next item's neighbour id (in flight / arrived)
    float q_n[3];
    {
        const int m0 = load_m(pc);
        float q0[3];
        load_q(pc, m0, q0);
        m_n = load_m(pn);
        finish(pc, m0, q0, a1, a2);
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const u32x2* rowp = sRow + h * 16;

    while (pc.ci < nb) {
        // the next item's coordinates and the one after's neighbour id go out now: a whole item of latency cover each
        const Pos pnn = advance(pn);
        load_q(pn, m_n, q_n);
        const int m_nn = load_m(pnn);

        const int b = b0 + bstride * pc.ci;
        const size_t pt = (size_t)b * N + pc.i;
        const int cl = pc.pass * CH + W * pc.qc + NC * l31;          // first of this lane's NC channels
        const unsigned colb = (unsigned)(C + cl) * ES;               // byte offset of this lane's columns in a fm row (support 0)
        unsigned off[KH];
#pragma unroll
        for (int r = 0; r < KH; ++r) off[r] = rowp[r][0] + colb;
        __amdgpu_buffer_rsrc_t frs, ars, wrs;
        if (!SURFACE)
            frs = __builtin_amdgcn_make_buffer_rsrc(rfm_uniform(fm + (size_t)b * N * fstride), 0,
                                                    (int)((size_t)N * fstride * ES), 0x00020000);
        ars = __builtin_amdgcn_make_buffer_rsrc(rfm_uniform(argrow + pt * SC), 0, SC * 2, 0x00020000);
        if (WF) wrs = __builtin_amdgcn_make_buffer_rsrc(rfm_uniform(fwin + pt * SC), 0, SC * (int)ES, 0x00020000);
        raw_t fcr = {};
        if (!SURFACE) fcr = *reinterpret_cast<const raw_t*>(fm + pt * fstride + cl);
        const float* pB1 = (h ? sDy : sDx) + cl;
        const float* pB2 = sDz + cl;                               // (upper half: multiplied by the zero half of a2)
        const unsigned acol = (unsigned)cl * 2u, wcol = (unsigned)cl * ES;

        raw_t f[2][KH];
        f32x16 acc[NC];
        auto gathers = [&](int s, raw_t* fs) {                      // the neighbours' support values of support s
            if (!SURFACE) {
#pragma unroll
                for (int r = 0; r < KH; ++r) fs[r] = Vec::gather(frs, off[r], (unsigned)(s * C) * ES);
            }
        };
        auto theta = [&](int s, int q) {                            // theta tile q of support s
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, pB1[s * C + q], zero16, 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, pB2[s * C + q], acc[q], 0, 0, 0);
        };
        float msum[NC];
        // support s: its gathers (fa) went out one tile ago, its theta tiles sit in acc; the next tile's gathers go out first, each
        // of its theta tiles as soon as this tile's products have left that accumulator (they run under the scan)
        auto tile = [&](int s, const raw_t* fa, raw_t* fb, bool more) {
            if (more) gathers(s + 1, fb);
            const unsigned so = (unsigned)(s * C);
            unsigned rid[NC];
            float wfv[NC];
            bool hiw[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                float p[KH];
#pragma unroll
                for (int r = 0; r < KH; ++r) {
                    if (SURFACE) p[r] = acc[q][r];
                    else {
                        const float fv = Vec::get(fa[r], q);
                        p[r] = mul_rn(RFM_RELU(acc[q][r]), fv);
                        if (WF) sF[(r * NC + q) * 64 + lane] = fv;       // for the winner's look-up (the LDS pipe idles; a select
                    }                                                    // per neighbour would cost the VALU as much as the compare)
                }
                if (more) theta(s + 1, q);
                float best = p[0];
#pragma unroll
                for (int r = 1; r < KH; ++r) best = fmaxf(best, p[r]);
                // SURFACE: first r whose CLAMPED theta equals the clamped maximum: theta_r >= thr, thr = 1 above 1, -inf at or below 0
                const float thr = SURFACE ? (best > 1.f ? 1.f : (best > 0.f ? best : -INFINITY)) : best;
                int arg = 0;
#pragma unroll
                for (int r = KH - 1; r >= 0; --r) arg = (SURFACE ? p[r] >= thr : p[r] == thr) ? r : arg;
                if (SURFACE) best = RFM_RELU(best);
                // the two halves of the neighbour list meet: the lower half (the earlier neighbours) wins ties
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                const float blo = __uint_as_float(sw[0]), bhi = __uint_as_float(sw[1]);
                hiw[q] = bhi > blo;
                const float bm = hiw[q] ? bhi : blo;
                rid[q] = rowp[arg][1];                                  // row id of this half's winner
                wfv[q] = WF ? sF[(arg * NC + q) * 64 + lane] : 0.f;     // ... and its support value
                msum[q] = s == 0 ? bm : add_rn(msum[q], bm);
            }
            // both halves learn the other's candidates, so that ONE half writes a lane's NC adjacent results with one store (a
            // 2-byte store per column from whichever half won leaves every 128-byte line to two interleaved partial writes:
            // measured 20x slower on the write-only surface layer)
            unsigned fin[NC / 2];
#pragma unroll
            for (int q = 0; q < NC; q += 2) {
                const unsigned mine2 = rid[q] | (rid[q + 1] << 16);
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(mine2, mine2, false, false);
                const unsigned msk = (hiw[q] ? 0x0000ffffu : 0u) | (hiw[q + 1] ? 0xffff0000u : 0u);
                fin[q >> 1] = (sw[1] & msk) | (sw[0] & ~msk);
            }
            if (WF) {
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(wfv[q]), __float_as_uint(wfv[q]), false, false);
                    wfv[q] = __uint_as_float(hiw[q] ? sw[1] : sw[0]);
                }
            }
            if (h == 0) {
                if constexpr (NC == 2) __builtin_amdgcn_raw_buffer_store_b32(fin[0], ars, (int)acol, (int)(so * 2u), 2);
                else __builtin_amdgcn_raw_buffer_store_b64(u32x2{fin[0], fin[NC / 2 - 1]}, ars, (int)acol, (int)(so * 2u), 2);
            } else if (WF) {
                Vec::store_nt(wrs, wfv, wcol, so * ES);
            }
        };
        gathers(0, f[0]);
#pragma unroll
        for (int q = 0; q < NC; ++q) theta(0, q);
        if (ST > 0) {
#pragma unroll
            for (int s = 0; s < ST; ++s) tile(s, f[s & 1], f[(s & 1) ^ 1], s + 1 < ST);
        } else {
            for (int s = 0; s < S; s += 2) {
                tile(s, f[0], f[1], s + 1 < S);
                if (s + 1 < S) tile(s + 1, f[1], f[0], s + 2 < S);
            }
        }
        float v[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            v[q] = __fdiv_rn(msum[q], (float)S);
            if (!SURFACE) v[q] = add_rn(Vec::get(fcr, q), v[q]);
        }
        if (h == 0) {
            const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(rfm_uniform(out + pt * C), 0, C * (int)ES, 0x00020000);
            Vec::store_nt(ors, v, wcol, 0);
        }
        // the next item's operands (its coordinates arrived long ago) and row table (this item's reads of it are done)
        finish(pn, m_n, q_n, a1, a2);
        pc = pn; pn = pnn; m_n = m_nn;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static size_t rfm_l2_budget() {
    static const size_t v = [] { const char* e = getenv("HSP_RF_L2_BYTES"); return e ? (size_t)atoll(e) : ((size_t)5 << 19); }();
    return v;                                                       // 2.5 MiB of a 4 MiB L2 for the gather working set
}

// passes over the channels: CPP = C / (W nsplit) must be 1, 2 or 4 (W = 32 NC columns per wave), and a pass's gather set should
// fit the L2 budget; 0 when no such split exists
static int rfm_nsplit(int N, int S, int C, size_t es, int W) {
    if (C % W) return 0;
    int ns = 1;
    while (C / (W * ns) > 4) ns *= 2;
    while (C / (W * ns) > 1 && (size_t)N * S * (C / ns) * es > rfm_l2_budget()) ns *= 2;
    const int cpp = C / (W * ns);
    if (C % (W * ns) || (cpp != 1 && cpp != 2 && cpp != 4)) return 0;
    return ns;
}

template <bool SURFACE, bool WF, typename FT, int NC, int KH, int ST>
static int rfm_launch2(const float* xyz, const int32_t* idx, const float* dirs, const FT* fm, int B, int N, int k, int S, int C,
                       int nsplit, FT* out, uint16_t* argrow, FT* fwin, hipStream_t st) {
    const int CPP = C / (32 * NC * nsplit);
    const size_t lds = (size_t)(3 * S * C) * 4 + (size_t)(RFM_THREADS / 64) * (32 * 8 + (WF ? 64 * KH * NC * 4 : 0));
    auto kern = rf_fwd_mfma_kernel<SURFACE, WF, FT, NC, KH, ST>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error(e); return HSP_ERR_LAUNCH; }
    }
    static const int bpc_env = [] { const char* e = getenv("HSP_RFM_BPC"); return e ? atoi(e) : 0; }();
    const long long items = (long long)B * nsplit * N * CPP / (RFM_THREADS / 64);
    const int grid = persistent_blocks(items, bpc_env > 0 ? bpc_env : 3);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(RFM_THREADS), lds, st, xyz, idx, dirs, fm, B, N, k, S, C, nsplit, out, argrow, fwin);
    return check_launch();
}

template <bool SURFACE, typename FT, int NC, int KH, bool FAST>
static int rfm_launch(const float* xyz, const int32_t* idx, const float* dirs, const FT* fm, int B, int N, int k, int S, int C,
                      int nsplit, FT* out, uint16_t* argrow, FT* fwin, hipStream_t st) {
    const bool wf = !SURFACE && fwin != nullptr;
    if constexpr (FAST) {                                            // the network's shapes: S = 7 unrolled
        if (S == 7) {
            if (wf) return rfm_launch2<SURFACE, !SURFACE, FT, NC, KH, 7>(xyz, idx, dirs, fm, B, N, k, S, C, nsplit, out, argrow, fwin, st);
            return rfm_launch2<SURFACE, false, FT, NC, KH, 7>(xyz, idx, dirs, fm, B, N, k, S, C, nsplit, out, argrow, fwin, st);
        }
    }
    if (wf) return rfm_launch2<SURFACE, !SURFACE, FT, NC, KH, 0>(xyz, idx, dirs, fm, B, N, k, S, C, nsplit, out, argrow, fwin, st);
    return rfm_launch2<SURFACE, false, FT, NC, KH, 0>(xyz, idx, dirs, fm, B, N, k, S, C, nsplit, out, argrow, fwin, st);
}

// HSP_OK / error, or 1 when the shape is not covered (the caller falls back on the VALU schedule)
template <bool SURFACE, typename FT>
int rf_fwd_mfma(const float* xyz, const int32_t* idx, const float* dirs, const FT* fm, int B, int N, int k, int S, int C,
                FT* out, uint16_t* argrow, FT* fwin, hspStream_t stream) {
    const char* env = getenv("HSP_RF_MFMA");                        // (read per call: the tests switch schedules in-process)
    const bool off = env && env[0] == '0';
    constexpr int NC = 2;
    if (off || k > 32 || k < 1) return 1;
    if ((size_t)N * (S + 1) * C * sizeof(FT) >= ((size_t)1 << 31)) return 1;        // 32-bit row offsets
    if ((size_t)3 * S * C * 4 > 96 * 1024) return 1;
    const int nsplit = rfm_nsplit(SURFACE ? 1 : N, S, C, sizeof(FT), 32 * NC);
    if (!nsplit) return 1;
    hipStream_t st = as_stream(stream);
    const int kh = (k + 1) / 2;
    if (kh <= 1) return rfm_launch<SURFACE, FT, NC, 1, false>(xyz, idx, dirs, fm, B, N, k, S, C, nsplit, out, argrow, fwin, st);
    if (kh <= 4) return rfm_launch<SURFACE, FT, NC, 4, true>(xyz, idx, dirs, fm, B, N, k, S, C, nsplit, out, argrow, fwin, st);
    if (kh <= 10) return rfm_launch<SURFACE, FT, NC, 10, true>(xyz, idx, dirs, fm, B, N, k, S, C, nsplit, out, argrow, fwin, st);
    return rfm_launch<SURFACE, FT, NC, 16, false>(xyz, idx, dirs, fm, B, N, k, S, C, nsplit, out, argrow, fwin, st);
}

template int rf_fwd_mfma<true, float>(const float*, const int32_t*, const float*, const float*, int, int, int, int, int, float*,
                                      uint16_t*, float*, hspStream_t);
template int rf_fwd_mfma<false, float>(const float*, const int32_t*, const float*, const float*, int, int, int, int, int, float*,
                                       uint16_t*, float*, hspStream_t);
template int rf_fwd_mfma<true, bf16_t>(const float*, const int32_t*, const float*, const bf16_t*, int, int, int, int, int, bf16_t*,
                                       uint16_t*, bf16_t*, hspStream_t);
template int rf_fwd_mfma<false, bf16_t>(const float*, const int32_t*, const float*, const bf16_t*, int, int, int, int, int, bf16_t*,
                                        uint16_t*, bf16_t*, hspStream_t);

}  // namespace hsp
