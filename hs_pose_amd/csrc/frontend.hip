// frontend.hip -- the steps either side of the network at inference for gfx950.
//
//  * depth -> point cloud: replaces the device work of PC_sample (reference
//    network/point_sample/pc_sample.py:8-77): fuse = mask * (depth > 0); back-project the masked
//    pixels with the intrinsics; keep `samplenum` of them chosen by the HOST (np.random.choice, same RNG
//    consumption as the reference); / 1000.  Stage 1 compacts the masked pixel ids of every image in
//    row-major order (the order of torch's boolean indexing) and returns the counts the host needs to
//    draw; stage 2 back-projects only the chosen pixels.  The reference materialises three H x W maps and
//    a variable-length (L,3) tensor per image and synchronises once per image; here: one sync per batch.
//  * (R|t) assembly: replaces generate_RT(..., mode='vec') (tools/geom_utils.py:232-244 with
//    tools/rot_utils.py:39-100): confidence-weighted orthogonalisation of the two predicted axes and
//    the 4x4 pose matrix, one lane per object instead of ~40 tiny launches.
#include "common.h"

namespace hsp {

#define PC_CHUNK 4096          // pixels per workgroup: 256 threads x 16 consecutive pixels (row-major order is kept)

__device__ __forceinline__ bool pc_valid(float m, float d) { return m * (d > 0.f ? 1.f : 0.f) > 0.f; }

// row-major stream compaction of {p : mask[p] > 0 && depth[p] > 0}, many workgroups per image, two launches:
//   pc_count_kernel   grid (nchunk, B): valid pixels of each 4096-pixel chunk -> cnt[b][chunk]
//   pc_write_kernel   grid (nchunk, B): offset = sum of the earlier chunks' counts, scan inside the chunk, ids written;
//                                       the last chunk's workgroup writes the image's total
// (one workgroup per image moved 75 GB/s: 16 workgroups on 256 CUs)
__global__ __launch_bounds__(256) void pc_count_kernel(const float* __restrict__ mask, const float* __restrict__ depth,
                                                       int HW, int nchunk, int32_t* __restrict__ cnt) {
    __shared__ int wsum[4];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const float* mb = mask + (size_t)b * HW;
    const float* db = depth + (size_t)b * HW;
    const int lo = min(chunk * PC_CHUNK + tid * 16, HW), hi = min(lo + 16, HW);
    int c = 0;
    for (int p = lo; p < hi; ++p) c += pc_valid(mb[p], db[p]) ? 1 : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
    if ((tid & 63) == 0) wsum[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) cnt[(size_t)b * nchunk + chunk] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void pc_write_kernel(const float* __restrict__ mask, const float* __restrict__ depth,
                                                       int HW, int nchunk, const int32_t* __restrict__ cnt,
                                                       int32_t* __restrict__ pix, int32_t* __restrict__ count) {
    __shared__ int red[4];
    __shared__ int wpre[4];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    // offset of this chunk = sum of the counts of the image's earlier chunks
    int part = 0;
    for (int c = tid; c < chunk; c += 256) part += cnt[(size_t)b * nchunk + c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m);
    if (lane == 0) red[wv] = part;
    const float* mb = mask + (size_t)b * HW;
    const float* db = depth + (size_t)b * HW;
    const int lo = min(chunk * PC_CHUNK + tid * 16, HW), hi = min(lo + 16, HW);
    unsigned bits = 0;
    for (int p = lo; p < hi; ++p) bits |= (pc_valid(mb[p], db[p]) ? 1u : 0u) << (p - lo);
    const int c = __popc(bits);
    int incl = c;                                            // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wpre[wv] = incl;
    __syncthreads();
    const int base = (red[0] + red[1]) + (red[2] + red[3]);
    int woff = 0;
    for (int w = 0; w < wv; ++w) woff += wpre[w];
    int off = base + woff + incl - c;
    int32_t* out = pix + (size_t)b * HW;
    for (int p = lo; p < hi; ++p)
        if ((bits >> (p - lo)) & 1u) out[off++] = p;
    if (chunk == nchunk - 1 && tid == 255) count[b] = base + woff + incl;
}

// PC[b,s,:] = ( (u - cx) * d / fx, (v - cy) * d / fy, d ) / 1000   for pixel pix[b, choose[b,s]]
__global__ __launch_bounds__(256) void pc_gather_kernel(const float* __restrict__ depth,
                                                        const float* __restrict__ coor2d,
                                                        const float* __restrict__ camK,
                                                        const int32_t* __restrict__ pix,
                                                        const int32_t* __restrict__ choose, int B, int HW, int S,
                                                        float* __restrict__ pc) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * S) return;
    const int b = e / S;
    const int p = pix[(size_t)b * HW + choose[e]];
    const float d = depth[(size_t)b * HW + p];
    const float u = coor2d[((size_t)b * 2 + 0) * HW + p], v = coor2d[((size_t)b * 2 + 1) * HW + p];
    const float* K = camK + (size_t)b * 9;
    const float fx = K[0], fy = K[4], ux = K[2], uy = K[5];
    const float x = __fdiv_rn(mul_rn(sub_rn(u, ux), d), fx);
    const float y = __fdiv_rn(mul_rn(sub_rn(v, uy), d), fy);
    pc[(size_t)e * 3 + 0] = __fdiv_rn(x, 1000.0f);
    pc[(size_t)e * 3 + 1] = __fdiv_rn(y, 1000.0f);
    pc[(size_t)e * 3 + 2] = __fdiv_rn(d, 1000.0f);
}

// dataset-side variant (datasets/load_data.py:322-333 then :275): numpy promotes to float64 --
// ((u - cx) * d / fx evaluated in double with a double K), rounds to fp32, then / 1000 in fp32.
__global__ __launch_bounds__(256) void depth_to_pcl_kernel(const float* __restrict__ depth,
                                                           const float* __restrict__ xymap,
                                                           const double* __restrict__ camK,
                                                           const int32_t* __restrict__ pix,
                                                           const int32_t* __restrict__ choose, int B, int HW, int S,
                                                           float* __restrict__ pc) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * S) return;
    const int b = e / S;
    const int p = pix[(size_t)b * HW + choose[e]];
    const double d = (double)depth[(size_t)b * HW + p];
    const double u = (double)xymap[((size_t)b * 2 + 0) * HW + p], v = (double)xymap[((size_t)b * 2 + 1) * HW + p];
    const double* K = camK + (size_t)b * 9;
    const double x = __ddiv_rn(__dmul_rn(__dsub_rn(u, K[2]), d), K[0]);
    const double y = __ddiv_rn(__dmul_rn(__dsub_rn(v, K[5]), d), K[4]);
    pc[(size_t)e * 3 + 0] = __fdiv_rn((float)x, 1000.0f);
    pc[(size_t)e * 3 + 1] = __fdiv_rn((float)y, 1000.0f);
    pc[(size_t)e * 3 + 2] = __fdiv_rn((float)d, 1000.0f);
}

__device__ __forceinline__ void rodrigues_apply(const float rx[3], float s, float c, const float v[3], float o[3]) {
    // rows of to_rot_matrix_in_batch (rot_utils.py:67-75) times v
    const float t = 1.f - c;
    o[0] = (rx[0] * rx[0] * t + c) * v[0] + (rx[0] * rx[1] * t - rx[2] * s) * v[1] + (rx[0] * rx[2] * t + rx[1] * s) * v[2];
    o[1] = (rx[1] * rx[0] * t + rx[2] * s) * v[0] + (rx[1] * rx[1] * t + c) * v[1] + (rx[1] * rx[2] * t - rx[0] * s) * v[2];
    o[2] = (rx[0] * rx[2] * t - rx[1] * s) * v[0] + (rx[2] * rx[1] * t + rx[0] * s) * v[1] + (rx[2] * rx[2] * t + c) * v[2];
}

__device__ __forceinline__ void normalize3(float v[3]) {     // F.normalize: v / max(|v|, 1e-12)
    const float n = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    v[0] /= n; v[1] /= n; v[2] /= n;
}

// generate_RT(mode='vec'): f_red := 0 where sym[:,0]==1; (new_y,new_x) = get_vertical_rot_vec_in_batch;
// R = get_rot_mat_y_first(new_y,new_x) = stack(x,y,z) as columns; res = [[R, T],[0,1]]
__global__ __launch_bounds__(64) void generate_rt_kernel(const float* __restrict__ p_green,
                                                         const float* __restrict__ p_red,
                                                         const float* __restrict__ f_green,
                                                         const float* __restrict__ f_red,
                                                         const float* __restrict__ T,
                                                         const float* __restrict__ sym, int sym_stride, int B,
                                                         float* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= B) return;
    const float c1 = f_green[i];
    const float c2 = (sym[(size_t)i * sym_stride] == 1.0f) ? 0.f : f_red[i];
    float y[3] = {p_green[i * 3], p_green[i * 3 + 1], p_green[i * 3 + 2]};
    float z[3] = {p_red[i * 3], p_red[i * 3 + 1], p_red[i * 3 + 2]};
    float rx[3] = {y[1] * z[2] - y[2] * z[1], y[2] * z[0] - y[0] * z[2], y[0] * z[1] - y[1] * z[0]};
    const float rn = sqrtf(rx[0] * rx[0] + rx[1] * rx[1] + rx[2] * rx[2]) + 1e-8f;
    rx[0] /= rn; rx[1] /= rn; rx[2] /= rn;
    float cs = y[0] * z[0] + y[1] * z[1] + y[2] * z[2];
    cs = fminf(fmaxf(cs, -1.f + 1e-6f), 1.f - 1e-6f);
    const float theta = acosf(cs);
    const float half_pi = 1.57079632679489661923f;
    const float th2 = c1 / (c1 + c2) * (theta - half_pi);
    const float th1 = c2 / (c1 + c2) * (theta - half_pi);
    float ny[3], nz[3];
    rodrigues_apply(rx, sinf(th1), cosf(th1), y, ny);
    rodrigues_apply(rx, sinf(-th2), cosf(-th2), z, nz);
    // get_rot_mat_y_first(y = ny, x = nz)
    normalize3(ny);
    float zz[3] = {nz[1] * ny[2] - nz[2] * ny[1], nz[2] * ny[0] - nz[0] * ny[2], nz[0] * ny[1] - nz[1] * ny[0]};   // cross(x, y)
    normalize3(zz);
    const float xx[3] = {ny[1] * zz[2] - ny[2] * zz[1], ny[2] * zz[0] - ny[0] * zz[2], ny[0] * zz[1] - ny[1] * zz[0]};   // cross(y, z)
    float* o = out + (size_t)i * 16;
    for (int r = 0; r < 3; ++r) {
        o[r * 4 + 0] = xx[r]; o[r * 4 + 1] = ny[r]; o[r * 4 + 2] = zz[r]; o[r * 4 + 3] = T[i * 3 + r];
    }
    o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}

}  // namespace hsp

using namespace hsp;

extern "C" size_t hsp_pc_compact_workspace_bytes(int B, int HW) {
    if (B <= 0 || HW <= 0) return 0;
    return (size_t)B * ((HW + PC_CHUNK - 1) / PC_CHUNK) * sizeof(int32_t);
}

extern "C" int hsp_pc_compact(const float* mask, const float* depth, int B, int HW, int32_t* pix, int32_t* count,
                              void* ws, size_t ws_bytes, hspStream_t stream) {
    if (!mask || !depth || !pix || !count || B <= 0 || HW <= 0) return HSP_ERR_BAD_ARG;
    if (!ws || ws_bytes < hsp_pc_compact_workspace_bytes(B, HW)) return HSP_ERR_WORKSPACE;
    const int nchunk = (HW + PC_CHUNK - 1) / PC_CHUNK;
    if (nchunk > 65535) return HSP_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    int32_t* cnt = reinterpret_cast<int32_t*>(ws);
    hipLaunchKernelGGL(pc_count_kernel, dim3(nchunk, B), dim3(256), 0, st, mask, depth, HW, nchunk, cnt);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(pc_write_kernel, dim3(nchunk, B), dim3(256), 0, st, mask, depth, HW, nchunk, cnt, pix, count);
    return check_launch();
}

extern "C" int hsp_pc_gather(const float* depth, const float* coor2d, const float* camK, const int32_t* pix,
                             const int32_t* choose, int B, int HW, int S, float* pc, hspStream_t stream) {
    if (!depth || !coor2d || !camK || !pix || !choose || !pc || B <= 0 || HW <= 0 || S <= 0) return HSP_ERR_BAD_ARG;
    hipLaunchKernelGGL(pc_gather_kernel, dim3((B * S + 255) / 256), dim3(256), 0, as_stream(stream), depth, coor2d, camK,
                       pix, choose, B, HW, S, pc);
    return check_launch();
}

extern "C" int hsp_depth_to_pcl(const float* depth, const float* xymap, const double* camK, const int32_t* pix,
                                const int32_t* choose, int B, int HW, int S, float* pc, hspStream_t stream) {
    if (!depth || !xymap || !camK || !pix || !choose || !pc || B <= 0 || HW <= 0 || S <= 0) return HSP_ERR_BAD_ARG;
    hipLaunchKernelGGL(depth_to_pcl_kernel, dim3((B * S + 255) / 256), dim3(256), 0, as_stream(stream), depth, xymap,
                       camK, pix, choose, B, HW, S, pc);
    return check_launch();
}

extern "C" int hsp_generate_rt(const float* p_green, const float* p_red, const float* f_green, const float* f_red,
                               const float* T, const float* sym, int sym_stride, int B, float* out,
                               hspStream_t stream) {
    if (!p_green || !p_red || !f_green || !f_red || !T || !sym || !out || B <= 0 || sym_stride <= 0) return HSP_ERR_BAD_ARG;
    hipLaunchKernelGGL(generate_rt_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), p_green, p_red, f_green,
                       f_red, T, sym, sym_stride, B, out);
    return check_launch();
}
