// common.h -- shared helpers for the libhsp.so HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <math.h>

#include "hsp.h"

#define HSP_WAVE 64
#define HSP_NUM_XCD 8            // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only)
#define HSP_NUM_CU 256

namespace hsp {

void set_last_hip_error(hipError_t e);

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return HSP_ERR_LAUNCH;
    }
    return HSP_OK;
}

inline hipStream_t as_stream(hspStream_t s) { return reinterpret_cast<hipStream_t>(s); }

// knn.hip -> knn_exact.hip: the xyz search by (distance, index) plus per-row flags: bit 0 "two of the k + drop + 1 nearest are equally
// far", bit 1 the same for the k2 + drop + 1 nearest
// (idx2 (B,N,k2), may be null: the first k2 entries of every list again -- the short list of every unflagged row)
// knn.hip -> knn_exact.hip: the feature-space search (any C != 3) with a flag byte per row "two of the k + drop + 1 nearest are equally
// far"; ws / ws_bytes as for hsp_knn_f32 (|x|^2 per row comes first in it); dmat (may be null): the (B, N, N) distances as
// [candidate][query], left for the tie pass
int knn_feat_select_flags(const float* x, int B, int N, int C, int k, int drop, int quad_mode, int32_t* idx, void* ws, size_t ws_bytes,
                          uint8_t* tie, float* dmat, hspStream_t stream);
// *needs_tie_pass: false when the selection replayed its flagged rows itself (then tie is not written)
int knn3_select_flags(const float* x, int B, int N, int k, int drop, int k2, int32_t* idx, int32_t* idx2, uint8_t* tie,
                      hipStream_t st, bool* needs_tie_pass);

// persistent grid: a multiple of the XCD count so that block % 8 == XCD for every block
inline int persistent_blocks(long long work_items, int blocks_per_cu) {
    long long g = (long long)HSP_NUM_CU * blocks_per_cu;
    if (work_items < g) g = ((work_items + HSP_NUM_XCD - 1) / HSP_NUM_XCD) * HSP_NUM_XCD;
    if (g < HSP_NUM_XCD) g = HSP_NUM_XCD;
    return (int)g;
}

// exact fp32 helpers that the compiler may not contract into fma
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }

// |v|^2 of an xyz row in ATen's order for C == 3: (x*x + y*y) + z*z, products rounded separately
__device__ __forceinline__ float quad3(float x, float y, float z) {
    return add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z));
}

// k-ordered fma chain == torch.bmm (MKL sgemm) for K = 3
__device__ __forceinline__ float dot3_chain(float ax, float ay, float az, float bx, float by, float bz) {
    return __fmaf_rn(az, bz, __fmaf_rn(ay, by, mul_rn(ax, bx)));
}

// sum of the squares of three numbers the way ATen's vector_norm accumulates them on the CPU: acc = fma(x, x, acc) in element
// order (its norm kernels are compiled with fma contraction) -- F.normalize over a dimension of size 3, contiguous (dim = -1:
// the neighbour directions) or strided (dim = 0: the support directions), bit for bit (oracle/gen_golden_exact.py)
__device__ __forceinline__ float norm2_chain(float x, float y, float z) { return __fmaf_rn(z, z, __fmaf_rn(y, y, mul_rn(x, x))); }

// unit vector from p to q the way F.normalize does it: v / max(|v|, 1e-12), |v| = sqrt(sum of squares)
__device__ __forceinline__ float3 unit_dir(float px, float py, float pz, float qx, float qy, float qz) {
    float dx = sub_rn(qx, px), dy = sub_rn(qy, py), dz = sub_rn(qz, pz);
    float n2 = norm2_chain(dx, dy, dz);
    float nrm = fmaxf(sqrtf(n2), 1e-12f);              // (sqrtf: correctly rounded; __fsqrt_rn is the 1-ulp native one)
    return make_float3(__fdiv_rn(dx, nrm), __fdiv_rn(dy, nrm), __fdiv_rn(dz, nrm));
}

// the same direction for gradient work: one v_rsq_f32 and three multiplies instead of a correctly rounded sqrt and
// three divisions (~45 instructions); agrees with unit_dir to ~1 ulp, which only ever scales a gradient term
__device__ __forceinline__ float3 unit_dir_fast(float px, float py, float pz, float qx, float qy, float qz) {
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    const float inv = fminf(__builtin_amdgcn_rsqf(dx * dx + dy * dy + dz * dz), 1e12f);   // 1 / max(|v|, 1e-12)
    return make_float3(dx * inv, dy * inv, dz * inv);
}

// ------------------------------------------------------------------------------------------------
// instruction-rate notes for gfx950 (tools/ubench/valu_rate.hip; clocks per wave64 instruction and SIMD, >= 2 waves resident):
// v_mul / v_fmac / v_mov (VGPR source) / v_and / v_add / shifts ~2.3-2.8; v_max / v_min (f32 and i32), v_med3, v_cmp,
// v_cndmask, anything with an SGPR or inline-constant operand ~4.2 ALONE -- but a stream that alternates them with the first
// class runs at ~2.1 per instruction (max + mul pairs: 4.2 per pair), so a mixed loop prices every VALU instruction at ~2.1-2.5.
// Measured and rejected in the receptive-field forward: running (max, arg, payload) updates as one compare + moves under a
// narrowed EXEC (s_and_saveexec / v_cmpx) instead of selects: 106 vs 96 us at B16 N1028 C128 -- the asm blocks pin the schedule.
// ------------------------------------------------------------------------------------------------
// feature storage type of a kernel: fp32, or bfloat16 bits (BASELINE configs[3]: features / fm / gradients stored in
// bf16, every kernel still computes in fp32 -- loads widen, stores round to nearest even).  xyz, support directions
// and theta never go through these (gcn3d.py:57,59 keeps them fp32).
// ------------------------------------------------------------------------------------------------
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf16_bits_to_f32(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ unsigned f32_to_bf16_bits(float f) {            // round to nearest even; NaN stays NaN
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

template <typename FT> struct Feat;
template <> struct Feat<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ void st_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }
    static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ void st4_nt(float* p, float4 v) {
        __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p + 1);
        __builtin_nontemporal_store(v.z, p + 2); __builtin_nontemporal_store(v.w, p + 3);
    }
    static __device__ __forceinline__ float rnd(float v) { return v; }        // value as it will read back from storage
};
template <> struct Feat<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_bits_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)f32_to_bf16_bits(v); }
    static __device__ __forceinline__ void st_nt(bf16_t* p, float v) { __builtin_nontemporal_store((bf16_t)f32_to_bf16_bits(v), p); }
    static __device__ __forceinline__ float4 ld4(const bf16_t* p) {                                  // 8-byte aligned
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    }
    static __device__ __forceinline__ uint2 pack4(float4 v) {
        return make_uint2(f32_to_bf16_bits(v.x) | (f32_to_bf16_bits(v.y) << 16), f32_to_bf16_bits(v.z) | (f32_to_bf16_bits(v.w) << 16));
    }
    static __device__ __forceinline__ void st4(bf16_t* p, float4 v) { *reinterpret_cast<uint2*>(p) = pack4(v); }
    static __device__ __forceinline__ void st4_nt(bf16_t* p, float4 v) {
        const uint2 u = pack4(v);
        unsigned* q = reinterpret_cast<unsigned*>(p);
        __builtin_nontemporal_store(u.x, q); __builtin_nontemporal_store(u.y, q + 1);
    }
    static __device__ __forceinline__ float rnd(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
};

}  // namespace hsp
