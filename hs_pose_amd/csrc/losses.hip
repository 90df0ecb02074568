// losses.hip -- the 19 training-loss terms of the PoseNet_only stage and their gradients in five launches
// (SURVEY 8 f-1).  Replaces, for the device path, the ~2 200 ATen kernels of
//   losses/fs_net_loss.py:95-235   losses/recon_loss.py:464-649   losses/geometry_loss.py:123-150
//   losses/prop_loss.py:156-276    tools/rot_utils.py:39-86       tools/plane_utils.py:24-49
// wired as network/HSPose.py:84-160 wires them (the axis confidences are constants everywhere but in R_con; the face
// confidences are constants in the plane fits).
//
// Shape of the computation.  Everything that is per point is a sum over the points of a cloud of terms whose only
// non-point inputs are a few numbers per cloud; everything else is a few hundred flops per cloud:
//   forward    prep (per cloud: the predicted frames the point terms need)
//              points pass (per cloud: loss sums, the sign sums that ARE the gradients w.r.t. the per-cloud inputs,
//                           and the 6 x 9 weighted moment sums of the plane fits, in fp64)
//              finish (per cloud: plane fits and the box terms; then the 19 sums over the batch, in index order)
//   backward   per-cloud program again, in forward-mode automatic differentiation with ONE TANGENT DIRECTION PER LANE
//              (68 inputs per cloud: 54 moment sums + the 14 pose numbers): every lane runs the same scalar program,
//              lane l carries d/d(input l) -- the derivative of the whole small program falls out with no hand-derived
//              adjoint; the point sums enter through their (constant) sign sums
//              points pass 2 (per point: the closed-form gradients of the per-point terms + the moment-sum gradients)
// All sums are taken in a fixed order (one workgroup per cloud, fixed tree): results are bit-reproducible run to run.
#include "common.h"

namespace hsp {
namespace {

constexpr int NT = HSP_LOSS_TERMS;          // 19
enum Term { T_ROT1, T_ROT1_COS, T_ROT2, T_ROT2_COS, T_ROT_R_A, T_TRAN, T_SIZE, T_R_CON, T_PER_P, T_P_F, T_VOTE, T_BB_R,
            T_BB_T, T_BB_S, T_BB_SELF, T_GEO, T_PM, T_SYM_RECON, T_SYM_RT };

// reductions of the points pass, per cloud (floats)
enum Red { R_RN = 0, R_RD = 6, R_RC = 12, R_LY = 18, R_LX = 19, R_SY = 20, R_CY = 23, R_SX = 24, R_CX = 27, R_LPM = 28,
           R_GPM = 29, R_CS = 38, R_LREC = 41, R_LRT = 42, R_GVEC = 43, R_GT = 46, NRED = 49 };
constexpr int NMOM = 54;                    // 6 faces x {wxx, wxy, wx, wyy, wy, w, wxz, wyz, wz}

// per-cloud frames written by prep for the point passes
enum Prm { P_PR = 0 /* 9: frame of Prop_pm, [i][j] */, P_NM = 9 /* 3: mirror-plane normal */, NPRM = 12 };

// network face order (y+, x+, z+, x-, z-, y-) -> loss order j = (x+, y+, z+, x-, y-, z-): network index of face j
__device__ __constant__ int kFacePerm[6] = {1, 0, 2, 3, 5, 4};

// ---- scalar type of the per-cloud program: float, or value + one tangent ---------------------------------------------
struct Dual {
    float v, d;
};
__device__ __forceinline__ Dual mk(float v, float d = 0.f) { return Dual{v, d}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    const float q = a.v / b.v;
    return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator+(Dual a, float b) { return {a.v + b, a.d}; }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return {a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return {a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, float b) { return {a.v / b, a.d / b}; }

__device__ __forceinline__ float val(float a) { return a; }
__device__ __forceinline__ float val(Dual a) { return a.v; }
__device__ __forceinline__ float tan_of(float) { return 0.f; }
__device__ __forceinline__ float tan_of(Dual a) { return a.d; }
template <class T> __device__ __forceinline__ T lift(float v);
template <> __device__ __forceinline__ float lift<float>(float v) { return v; }
template <> __device__ __forceinline__ Dual lift<Dual>(float v) { return {v, 0.f}; }

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }      // torch.sign
__device__ __forceinline__ float t_abs(float a) { return fabsf(a); }
__device__ __forceinline__ Dual t_abs(Dual a) { return {fabsf(a.v), sgnf(a.v) * a.d}; }
__device__ __forceinline__ float t_sqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ Dual t_sqrt(Dual a) {
    const float s = sqrtf(a.v);
    return {s, s > 0.f ? a.d / (2.f * s) : 0.f};
}
__device__ __forceinline__ float t_sin(float a) { return sinf(a); }
__device__ __forceinline__ Dual t_sin(Dual a) { return {sinf(a.v), cosf(a.v) * a.d}; }
__device__ __forceinline__ float t_cos(float a) { return cosf(a); }
__device__ __forceinline__ Dual t_cos(Dual a) { return {cosf(a.v), -sinf(a.v) * a.d}; }
__device__ __forceinline__ float t_exp(float a) { return expf(a); }
__device__ __forceinline__ Dual t_exp(Dual a) {
    const float e = expf(a.v);
    return {e, e * a.d};
}
__device__ __forceinline__ float t_acos(float a) { return acosf(a); }
__device__ __forceinline__ Dual t_acos(Dual a) { return {acosf(a.v), -a.d / sqrtf(1.f - a.v * a.v)}; }
// torch.clamp: the gradient passes where lo <= x <= hi
__device__ __forceinline__ float t_clamp(float a, float lo, float hi) { return fminf(fmaxf(a, lo), hi); }
__device__ __forceinline__ Dual t_clamp(Dual a, float lo, float hi) {
    return {fminf(fmaxf(a.v, lo), hi), (a.v >= lo && a.v <= hi) ? a.d : 0.f};
}
template <class T> __device__ __forceinline__ T t_sel(bool c, T a, T b) { return c ? a : b; }

template <class T> __device__ __forceinline__ T dot3(const T a[3], const T b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> __device__ __forceinline__ void cross3(const T a[3], const T b[3], T o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// torch.norm(dim=-1): sqrt of the sum of squares; its gradient at the origin is taken as 0 (t_sqrt)
template <class T> __device__ __forceinline__ T norm3(const T a[3]) { return t_sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
// F.normalize(p=2): v / max(|v|, 1e-12)
template <class T> __device__ __forceinline__ void normalize3(const T a[3], T o[3]) {
    T n = norm3(a);
    if (val(n) < 1e-12f) n = lift<T>(1e-12f);
    for (int i = 0; i < 3; ++i) o[i] = a[i] / n;
}

// nn.L1Loss / nn.SmoothL1Loss(beta) of one element
template <class T> __device__ __forceinline__ T elem_loss(T x, int smooth, float beta) {
    if (!smooth) return t_abs(x);
    return fabsf(val(x)) < beta ? (x * x) * (0.5f / beta) : t_abs(x) - 0.5f * beta;
}

// get_vertical_rot_vec_in_batch (tools/rot_utils.py:39-65): the two axes turned about their common normal by
// confidence-weighted shares of (angle - 90 deg).  c1, c2 are constants (detached confidences).
template <class T>
__device__ void vertical_axes(float c1, float c2, const T y[3], const T z[3], T ny[3], T nz[3]) {
    T ax[3];
    cross3(y, z, ax);
    const T an = norm3(ax) + 1e-8f;
    for (int i = 0; i < 3; ++i) ax[i] = ax[i] / an;
    const T theta = t_acos(t_clamp(dot3(y, z), -1.f + 1e-6f, 1.f - 1e-6f));
    const T excess = theta - 1.5707963267948966f;
    const T th_y = excess * (c2 / (c1 + c2));
    const T th_z = excess * (c1 / (c1 + c2));
    // Rodrigues (rot_utils.py:67-75): R = (k k^T)(1 - c) + c I + s [k]x, then R v
    auto rotate = [&](T s, T c, const T v[3], T o[3]) {
        const T t = 1.f - c;
        T Rm[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rm[i][j] = (ax[i] * ax[j]) * t + (i == j ? c : lift<T>(0.f));
        Rm[0][1] = Rm[0][1] + (-ax[2]) * s;
        Rm[0][2] = Rm[0][2] + ax[1] * s;
        Rm[1][0] = Rm[1][0] + ax[2] * s;
        Rm[1][2] = Rm[1][2] + (-ax[0]) * s;
        Rm[2][0] = Rm[2][0] + (-ax[1]) * s;
        Rm[2][1] = Rm[2][1] + ax[0] * s;
        for (int i = 0; i < 3; ++i) o[i] = Rm[i][0] * v[0] + Rm[i][1] * v[1] + Rm[i][2] * v[2];
    };
    rotate(t_sin(th_y), t_cos(th_y), y, ny);
    rotate(t_sin(-th_z), t_cos(-th_z), z, nz);
}

// get_rot_mat_y_first (rot_utils.py:77-86): columns (x, y, z), y kept.  pR[i][j]
template <class T> __device__ void rot_mat_y_first(const T y_in[3], const T x_in[3], T pR[9]) {
    T y[3], z[3], c[3], x[3];
    normalize3(y_in, y);
    cross3(x_in, y, c);
    normalize3(c, z);
    cross3(y, z, x);
    for (int i = 0; i < 3; ++i) {
        pR[3 * i + 0] = x[i];
        pR[3 * i + 1] = y[i];
        pR[3 * i + 2] = z[i];
    }
}

// what a cloud brings that is not a network output
struct CloudGT {
    float R[9], t[3], s[3], ms[3], sym[4], obj;
    bool rot_sym, keep;            // sym[0] == 1 ; sym[0] == 0 (the red axis is defined)
    bool cls_y, cls_yx, cls_none, skip;
    bool axis_mask[3];
};
__device__ void load_gt(CloudGT& c, const float* gt_R, const float* gt_t, const float* gt_s, const float* mean_shape,
                        const float* sym, const float* obj_id, int b) {
    for (int i = 0; i < 9; ++i) c.R[i] = gt_R[b * 9 + i];
    for (int i = 0; i < 3; ++i) { c.t[i] = gt_t[b * 3 + i]; c.s[i] = gt_s[b * 3 + i]; c.ms[i] = mean_shape[b * 3 + i]; }
    for (int i = 0; i < 4; ++i) c.sym[i] = sym[b * 4 + i];
    c.obj = obj_id[b];
    c.rot_sym = c.sym[0] == 1.f;
    c.keep = c.sym[0] == 0.f;
    const float mirrors = c.sym[1] + c.sym[2] + c.sym[3];
    c.cls_y = c.rot_sym && mirrors > 0.f;
    c.cls_yx = !c.rot_sym && c.sym[1] == 1.f;
    c.cls_none = !c.rot_sym && c.sym[1] != 1.f;
    c.skip = c.rot_sym && mirrors == 0.f;
    c.axis_mask[0] = c.keep && c.obj != 5.f;
    c.axis_mask[1] = true;
    c.axis_mask[2] = c.keep;
}

// the frames the point terms are evaluated with (prop_loss.py:156-189, 258-276)
template <class T>
__device__ void cloud_frames(const CloudGT& c, const T g[3], const T r[3], float fg, float fr, T pR[9], T nm[3], T ny[3], T nx[3]) {
    // (ny, nx): the predicted axes made perpendicular (shared with the box-rotation term, recon_loss.py:640)
    vertical_axes(fg, fr, g, r, ny, nx);
    T ys[3], xs[3], gx[3];
    for (int i = 0; i < 3; ++i) gx[i] = lift<T>(c.R[3 * i + 0]);             // ground-truth x axis stands in for the red one
    vertical_axes(fg, 1e-5f, g, gx, ys, xs);
    T ysel[3], xsel[3];
    for (int i = 0; i < 3; ++i) { ysel[i] = c.rot_sym ? ys[i] : ny[i]; xsel[i] = c.rot_sym ? xs[i] : nx[i]; }
    rot_mat_y_first(ysel, xsel, pR);
    T cr[3];
    cross3(r, g, cr);
    const T cn = norm3(cr) + 1e-8f;
    for (int i = 0; i < 3; ++i) nm[i] = cr[i] / cn;
}

// ---- kernel 1: per-cloud frames ---------------------------------------------------------------------------------------
__global__ void loss_prep_kernel(const float* gt_R, const float* gt_t, const float* gt_s, const float* mean_shape,
                                 const float* sym, const float* obj_id, const float* p_green, const float* p_red,
                                 const float* f_green, const float* f_red, int B, float* prm) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    CloudGT c;
    load_gt(c, gt_R, gt_t, gt_s, mean_shape, sym, obj_id, b);
    float g[3], r[3], pR[9], nm[3], ny[3], nx[3];
    for (int i = 0; i < 3; ++i) { g[i] = p_green[b * 3 + i]; r[i] = p_red[b * 3 + i]; }
    cloud_frames<float>(c, g, r, f_green[b], f_red[b], pR, nm, ny, nx);
    for (int i = 0; i < 9; ++i) prm[b * NPRM + P_PR + i] = pR[i];
    for (int i = 0; i < 3; ++i) prm[b * NPRM + P_NM + i] = nm[i];
}

// ---- per-point geometry shared by the two point passes ----------------------------------------------------------------
struct PointCtx {
    float P[3], canon[3], c[3];     // point, R^T (P - t_gt), P - T_pred
};
__device__ __forceinline__ void point_ctx(PointCtx& q, const CloudGT& c, const float* T, const float* P) {
    for (int i = 0; i < 3; ++i) { q.P[i] = P[i]; q.c[i] = P[i] - T[i]; }
    const float d0 = P[0] - c.t[0], d1 = P[1] - c.t[1], d2 = P[2] - c.t[2];
    for (int j = 0; j < 3; ++j) q.canon[j] = d0 * c.R[j] + d1 * c.R[3 + j] + d2 * c.R[6 + j];
}
// target of the reconstruction term and the mirrored point of the rt term (prop_loss.py:258-276)
__device__ __forceinline__ void sym_targets(const PointCtx& q, const CloudGT& c, const float* g, const float* T, const float* nm,
                                            float target[3], float mirrored[3], float& u) {
    u = 0.f;
    if (c.cls_yx || c.cls_y) {
        float m[3] = {c.cls_yx ? q.canon[0] : -q.canon[0], q.canon[1], -q.canon[2]};      // (x, y, -z) | (-x, y, -z)
        for (int i = 0; i < 3; ++i) target[i] = (m[0] * c.R[3 * i] + m[1] * c.R[3 * i + 1] + m[2] * c.R[3 * i + 2]) + c.t[i];
    } else if (c.cls_none) {
        for (int i = 0; i < 3; ++i) target[i] = q.P[i];
    } else {
        for (int i = 0; i < 3; ++i) target[i] = 0.f;
    }
    if (c.cls_y) {
        const float cg = q.c[0] * g[0] + q.c[1] * g[1] + q.c[2] * g[2];
        for (int i = 0; i < 3; ++i) mirrored[i] = q.P[i] + 2.0f * (cg * g[i] - q.c[i]);
    } else if (c.cls_yx) {
        u = (q.P[0] * nm[0] + q.P[1] * nm[1] + q.P[2] * nm[2]) - (nm[0] * T[0] + nm[1] * T[1] + nm[2] * T[2]);
        const float dist = -u;
        for (int i = 0; i < 3; ++i) mirrored[i] = q.P[i] + (2.0f * dist) * nm[i];
    } else {
        for (int i = 0; i < 3; ++i) mirrored[i] = 0.f;
    }
}

template <typename V> __device__ __forceinline__ V wave_sum(V v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- kernel 2: points pass 1 -- one workgroup per cloud ---------------------------------------------------------------
constexpr int PTS_THREADS = 512;
__global__ __launch_bounds__(PTS_THREADS) void loss_points_kernel(
    const float* __restrict__ PC, const float* gt_R, const float* gt_t, const float* gt_s, const float* mean_shape,
    const float* sym, const float* obj_id, const float* __restrict__ recon, const float* __restrict__ face_normal,
    const float* __restrict__ face_dis, const float* __restrict__ face_f, const float* p_green, const float* p_red,
    const float* pred_T, const float* __restrict__ prm, int N, float* __restrict__ red, double* __restrict__ mom) {
    const int b = blockIdx.x;
    CloudGT c;
    load_gt(c, gt_R, gt_t, gt_s, mean_shape, sym, obj_id, b);
    float g[3], r[3], T[3], pR[9], nm[3], half[3];
    for (int i = 0; i < 3; ++i) {
        g[i] = p_green[b * 3 + i]; r[i] = p_red[b * 3 + i]; T[i] = pred_T[b * 3 + i];
        nm[i] = prm[b * NPRM + P_NM + i];
        half[i] = (c.s[i] + c.ms[i]) / 2.0f;
    }
    for (int i = 0; i < 9; ++i) pR[i] = prm[b * NPRM + P_PR + i];

    float acc[NRED];
    double macc[NMOM];
    for (int i = 0; i < NRED; ++i) acc[i] = 0.f;
    for (int i = 0; i < NMOM; ++i) macc[i] = 0.0;

    for (int n = threadIdx.x; n < N; n += PTS_THREADS) {
        const size_t pn = (size_t)b * N + n;
        PointCtx q;
        point_ctx(q, c, T, PC + pn * 3);
        // geometry_loss.py:123-150
        {
            const float vy = (q.c[0] * g[0] + q.c[1] * g[1] + q.c[2] * g[2]) - q.canon[1];
            acc[R_LY] += fabsf(vy);
            const float s = sgnf(vy);
            for (int i = 0; i < 3; ++i) acc[R_SY + i] += s * q.c[i];
            acc[R_CY] += s;
            if (c.keep) {
                const float vx = (q.c[0] * r[0] + q.c[1] * r[1] + q.c[2] * r[2]) - q.canon[0];
                acc[R_LX] += fabsf(vx);
                const float sx = sgnf(vx);
                for (int i = 0; i < 3; ++i) acc[R_SX + i] += sx * q.c[i];
                acc[R_CX] += sx;
            }
        }
        // prop_loss.py:156-189
        for (int j = 0; j < 3; ++j) {
            const float v = (q.c[0] * pR[j] + q.c[1] * pR[3 + j] + q.c[2] * pR[6 + j]) - q.canon[j];
            acc[R_LPM] += fabsf(v);
            const float s = sgnf(v);
            for (int i = 0; i < 3; ++i) acc[R_GPM + 3 * i + j] += s * q.c[i];
            acc[R_CS + j] += s;
        }
        // prop_loss.py:258-276
        {
            float target[3], mir[3], u;
            sym_targets(q, c, g, T, nm, target, mir, u);
            const float* re = recon + pn * 3;
            float sv[3];
            for (int i = 0; i < 3; ++i) {
                acc[R_LREC] += fabsf(target[i] - (c.skip ? 0.f : re[i]));
                const float d = mir[i] - ((c.cls_y || c.cls_yx) ? re[i] : 0.f);
                acc[R_LRT] += fabsf(d);
                sv[i] = sgnf(d);
            }
            if (c.cls_y) {
                const float sg = sv[0] * g[0] + sv[1] * g[1] + sv[2] * g[2];
                const float cg = q.c[0] * g[0] + q.c[1] * g[1] + q.c[2] * g[2];
                for (int i = 0; i < 3; ++i) {
                    acc[R_GVEC + i] += 2.0f * (sg * q.c[i] + cg * sv[i]);
                    acc[R_GT + i] += -2.0f * (sg * g[i] - sv[i]);
                }
            } else if (c.cls_yx) {
                const float sn = sv[0] * nm[0] + sv[1] * nm[1] + sv[2] * nm[2];
                for (int i = 0; i < 3; ++i) {
                    acc[R_GVEC + i] += -2.0f * (sn * q.c[i] + u * sv[i]);
                    acc[R_GT + i] += 2.0f * sn * nm[i];
                }
            }
        }
        // recon_loss.py:464-543 (per-point face terms) and the moment sums of the plane fits (plane_utils.py:24-49)
        for (int j = 0; j < 6; ++j) {
            const int a = j % 3, nj = kFacePerm[j];
            const float sign = j < 3 ? 1.f : -1.f;
            const float* fn = face_normal + (pn * 6 + nj) * 3;
            const float fd = face_dis[pn * 6 + nj], ff = face_f[pn * 6 + nj];
            const float ng[3] = {sign * c.R[a], sign * c.R[3 + a], sign * c.R[6 + a]};
            const float dg = half[a] - sign * q.canon[a];
            acc[R_RN + j] += 1.0f - (fn[0] * ng[0] + fn[1] * ng[1] + fn[2] * ng[2]);
            acc[R_RD + j] += fabsf(fd - dg);
            float v[3];
            for (int i = 0; i < 3; ++i) v[i] = fn[i] * fd - ng[i] * dg;
            const float err = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const float conf = expf(-303.5f * err * err);
            acc[R_RC + j] += fabsf(conf - ff);
            const double w = ff, x = q.P[0] + fd * fn[0], y = q.P[1] + fd * fn[1], z = q.P[2] + fd * fn[2];
            double* m = macc + j * 9;
            m[0] += w * x * x; m[1] += w * x * y; m[2] += w * x;
            m[3] += w * y * y; m[4] += w * y; m[5] += w;
            m[6] += w * x * z; m[7] += w * y * z; m[8] += w * z;
        }
    }
    // fixed-order tree: lanes (xor butterfly), then the 8 waves in index order
    __shared__ float sf[PTS_THREADS / 64][NRED];
    __shared__ double sd[PTS_THREADS / 64][NMOM];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = 0; i < NRED; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) sf[wave][i] = s;
    }
    for (int i = 0; i < NMOM; ++i) {
        const double s = wave_sum(macc[i]);
        if (lane == 0) sd[wave][i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NRED; i += PTS_THREADS) {
        float s = 0.f;
        for (int w = 0; w < PTS_THREADS / 64; ++w) s += sf[w][i];
        red[b * NRED + i] = s;
    }
    for (int i = threadIdx.x; i < NMOM; i += PTS_THREADS) {
        double s = 0.0;
        for (int w = 0; w < PTS_THREADS / 64; ++w) s += sd[w][i];
        mom[b * NMOM + i] = s;
    }
}

// ---- the per-cloud program: contribution of cloud b to each of the 19 terms -------------------------------------------
// (already divided by the batch size etc.: the terms are the sums of these over b).  In Dual form `lin[k]` receives, for
// the three terms that are point sums over frames, the tangent of the term through those frames (sign sums x frame
// tangents); their VALUES come from the point sums directly.
template <class T>
__device__ void cloud_program(const CloudGT& c, const HspLossCfg& cfg, int B, int N, float scale2 /* B / #kept or 1 */,
                              const float* red, const T S[NMOM], const T g[3], const T r[3], T fg, T fr, const T Tp[3],
                              const T sp[3], T term[NT], float* bad) {
    const float fB = (float)B, fN = (float)N;
    const int sm = cfg.smooth_l1;
    for (int k = 0; k < NT; ++k) term[k] = lift<T>(0.f);
    float gg[3], rg[3];
    for (int i = 0; i < 3; ++i) { gg[i] = c.R[3 * i + 1]; rg[i] = c.R[3 * i + 0]; }

    // ---- fs_net_loss.py:95-235
    {
        T l1 = lift<T>(0.f), l2 = lift<T>(0.f), lt = lift<T>(0.f), ls = lift<T>(0.f), dgg = lift<T>(0.f), drr = lift<T>(0.f);
        for (int i = 0; i < 3; ++i) {
            l1 = l1 + elem_loss(g[i] - gg[i], sm, 0.5f);
            l2 = l2 + elem_loss(r[i] - rg[i], sm, 0.5f);
            lt = lt + elem_loss(Tp[i] - c.t[i], sm, 0.5f);
            ls = ls + elem_loss(sp[i] - c.s[i], sm, 0.5f);
            dgg = dgg + g[i] * gg[i];
            drr = drr + r[i] * rg[i];
        }
        term[T_ROT1] = l1 * (cfg.rot_1_w / (3.f * fB));
        term[T_ROT1_COS] = ((1.0f - dgg) * 2.0f) * (cfg.rot_1_w / fB);
        term[T_TRAN] = lt * (cfg.tran_w / (3.f * fB));
        term[T_SIZE] = ls * (cfg.size_w / (3.f * fB));
        if (c.keep) {
            term[T_ROT2] = l2 * (cfg.rot_2_w * scale2 / (3.f * fB));
            term[T_ROT2_COS] = ((1.0f - drr) * 2.0f) * (cfg.rot_2_w * scale2 / fB);
            term[T_ROT_R_A] = t_abs(dot3(g, r)) * (cfg.rot_regular * scale2 / fB);
        }
        auto target = [&](const T p[3], const float q[3]) {
            T d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
            const T n = norm3(d);
            return t_exp((n * n) * -13.7f);
        };
        T rc = elem_loss(target(g, gg) - fg, sm, 0.5f);
        if (c.keep) rc = rc + elem_loss(target(r, rg) - fr, sm, 0.5f);
        term[T_R_CON] = rc * (cfg.r_con_w / fB);
    }

    // ---- recon_loss.py:464-543: the per-point face terms are point sums (constants of this program)
    {
        float resn = 0.f, resd = 0.f, resc = 0.f;
        for (int j = 0; j < 6; ++j) {
            const int a = j % 3;
            if (a == 1 || c.keep) resn += red[R_RN + j] / fN;
            if (c.axis_mask[a]) { resd += red[R_RD + j] / fN; resc += red[R_RC + j] / fN; }
        }
        term[T_PER_P] = lift<T>((cfg.recon_n_w * resn + cfg.recon_d_w * resd) / 6.f / fB);
        term[T_P_F] = lift<T>(cfg.recon_f_w * resc / 6.f / fB);
        term[T_GEO] = lift<T>(cfg.geo_p_w * (red[R_LY] / (fB * fN) + scale2 * red[R_LX] / (fB * fN)));
        term[T_PM] = lift<T>(cfg.prop_pm_w * red[R_LPM] / (3.f * fB * fN));
        term[T_SYM_RECON] = lift<T>(cfg.prop_sym_w * red[R_LREC] / (3.f * fB * fN));
        term[T_SYM_RT] = lift<T>(cfg.prop_sym_w * red[R_LRT] / (3.f * fB * fN));
    }

    // ---- frames; tangents of the three frame-dependent point sums
    T pR[9], nm[3], ny[3], nx[3];
    cloud_frames<T>(c, g, r, val(fg), val(fr), pR, nm, ny, nx);
    if constexpr (sizeof(T) == sizeof(Dual)) {
        const float inv_bn = 1.f / (fB * fN);
        // geo: v = (P - T).g - canon  ->  dv = c.dg - g.dT
        float d = 0.f;
        for (int i = 0; i < 3; ++i) {
            d += red[R_SY + i] * tan_of(g[i]) - red[R_CY] * val(g[i]) * tan_of(Tp[i]);
            d += scale2 * (red[R_SX + i] * tan_of(r[i]) - red[R_CX] * val(r[i]) * tan_of(Tp[i]));
        }
        term[T_GEO].d = cfg.geo_p_w * inv_bn * d;
        // pm: v_j = sum_i c_i pR[i][j] - canon_j
        d = 0.f;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                d += red[R_GPM + 3 * i + j] * tan_of(pR[3 * i + j]) - red[R_CS + j] * val(pR[3 * i + j]) * tan_of(Tp[i]);
        term[T_PM].d = cfg.prop_pm_w * inv_bn / 3.f * d;
        // rt: through g (180 deg about y) or through the mirror normal, and T
        d = 0.f;
        for (int i = 0; i < 3; ++i) {
            if (c.cls_y) d += red[R_GVEC + i] * tan_of(g[i]);
            if (c.cls_yx) d += red[R_GVEC + i] * tan_of(nm[i]);
            d += red[R_GT + i] * tan_of(Tp[i]);
        }
        term[T_SYM_RT].d = cfg.prop_sym_w * inv_bn / 3.f * d;
    }

    // ---- recon_loss.py:555-649: planes through the votes, the box they make
    T nrm[2][3][3], cc[2][3];                       // [up / down][axis][xyz], signed offset
    T vote = lift<T>(0.f);
    bool isbad = false;
    for (int h = 0; h < 2; ++h) {
        const float sign = h == 0 ? 1.f : -1.f;
        for (int a = 0; a < 3; ++a) {
            const T* m = S + (3 * h + a) * 9;
            // normal equations M X = rhs, X by cofactors (the device form of torch.inverse for 3x3)
            const T m00 = m[0], m01 = m[1], m02 = m[2], m11 = m[3], m12 = m[4], m22 = m[5];
            const T A = m11 * m22 - m12 * m12, Bc = m02 * m12 - m01 * m22, C = m01 * m12 - m02 * m11;
            const T D = m12 * m02 - m01 * m22, E = m00 * m22 - m02 * m02, F = m02 * m01 - m00 * m12;
            const T G = m01 * m12 - m11 * m02, H = m01 * m02 - m00 * m12, I = m00 * m11 - m01 * m01;
            const T det = m00 * A + m01 * D + m02 * G;
            const T X0 = (A * m[6] + Bc * m[7] + C * m[8]) / det;
            const T X1 = (D * m[6] + E * m[7] + F * m[8]) / det;
            const T X2 = (G * m[6] + H * m[7] + I * m[8]) / det;
            const T norm2 = X0 * X0 + X1 * X1 + 1.0f;
            T dn[3] = {(X0 * X2) / (norm2 + 1e-8f), (X1 * X2) / (norm2 + 1e-8f), (-X2) / (norm2 + 1e-8f)};
            const T dl = norm3(dn);
            T n[3] = {dn[0] / dl, dn[1] / dl, dn[2] / dl};
            T co = X2 / t_sqrt(norm2);
            const float ax[3] = {sign * c.R[a], sign * c.R[3 + a], sign * c.R[6 + a]};
            const bool flip = val(n[0]) * ax[0] + val(n[1]) * ax[1] + val(n[2]) * ax[2] < 0.f;
            if (flip) { for (int i = 0; i < 3; ++i) n[i] = -n[i]; co = -co; }
            for (int i = 0; i < 3; ++i) { nrm[h][a][i] = n[i]; isbad = isbad || isnan(val(n[i])); }
            cc[h][a] = co;
            isbad = isbad || isnan(val(co));
            // the fitted foot point against the true one
            const float re_s = c.s[a] + c.ms[a];
            float fc[3], dots = 0.f;
            for (int i = 0; i < 3; ++i) { fc[i] = c.t[i] + ax[i] * re_s / 2.0f; dots += ax[i] * fc[i]; }
            T e = lift<T>(0.f);
            for (int i = 0; i < 3; ++i) e = e + t_abs(dn[i] - ax[i] * (-dots));
            if (c.axis_mask[a]) vote = vote + e / 3.f;
        }
    }
    *bad = isbad ? 1.f : 0.f;
    term[T_VOTE] = vote * (cfg.recon_v_w / 6.f / fB);
    {
        T frame[3][3];
        for (int i = 0; i < 3; ++i) { frame[0][i] = nx[i]; frame[1][i] = ny[i]; }
        cross3(nx, ny, frame[2]);
        T rr = lift<T>(0.f), tt = lift<T>(0.f), ss = lift<T>(0.f), self = lift<T>(0.f);
        for (int a = 0; a < 3; ++a) {
            if (!c.axis_mask[a]) continue;
            T eu = lift<T>(0.f), ed = lift<T>(0.f), par = lift<T>(0.f);
            for (int i = 0; i < 3; ++i) {
                eu = eu + t_abs(nrm[0][a][i] - frame[a][i]);
                ed = ed + t_abs(nrm[1][a][i] + frame[a][i]);
                par = par + t_abs(nrm[0][a][i] + nrm[1][a][i]);
            }
            rr = rr + eu / 3.f + ed / 3.f;
            self = self + par / 3.f;
            const T du = t_abs(dot3(nrm[0][a], Tp) + cc[0][a]);
            const T dd = t_abs(dot3(nrm[1][a], Tp) + cc[1][a]);
            tt = tt + t_abs(dd - du);
            const T hs = (sp[a] + c.ms[a]) / 2.0f;
            ss = ss + t_abs(hs - du) + t_abs(hs - dd);
            if (a != 1) self = self + t_abs(dot3(nrm[0][1], nrm[0][a])) + t_abs(dot3(nrm[1][1], nrm[1][a]));
        }
        term[T_BB_R] = rr * (cfg.recon_bb_r_w / 6.f / fB);
        term[T_BB_T] = tt * (cfg.recon_bb_t_w / 6.f / fB);
        term[T_BB_S] = ss * (cfg.recon_bb_s_w / 6.f / fB);
        term[T_BB_SELF] = self * (cfg.recon_bb_self_w / 6.f / fB);
    }
}

__device__ float rescale_of(const float* sym, int B) {
    int kept = 0;
    for (int b = 0; b < B; ++b) kept += sym[b * 4] == 0.f;
    return kept > 0 ? (float)B / (float)kept : 1.f;
}

// ---- kernel 3: forward finish -- the 19 terms ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* gt_R, const float* gt_t, const float* gt_s,
                                                          const float* mean_shape, const float* sym, const float* obj_id,
                                                          const float* p_green, const float* p_red, const float* f_green,
                                                          const float* f_red, const float* pred_T, const float* pred_s,
                                                          const float* red, const double* mom, HspLossCfg cfg, int B, int N,
                                                          float* per_cloud /* (B, 20) */, float* terms) {
    const float scale2 = rescale_of(sym, B);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        CloudGT c;
        load_gt(c, gt_R, gt_t, gt_s, mean_shape, sym, obj_id, b);
        float g[3], r[3], T[3], s[3], S[NMOM], term[NT], bad;
        for (int i = 0; i < 3; ++i) { g[i] = p_green[b * 3 + i]; r[i] = p_red[b * 3 + i]; T[i] = pred_T[b * 3 + i]; s[i] = pred_s[b * 3 + i]; }
        for (int i = 0; i < NMOM; ++i) S[i] = (float)mom[b * NMOM + i];
        cloud_program<float>(c, cfg, B, N, scale2, red + b * NRED, S, g, r, f_green[b], f_red[b], T, s, term, &bad);
        for (int k = 0; k < NT; ++k) per_cloud[b * (NT + 1) + k] = term[k];
        per_cloud[b * (NT + 1) + NT] = bad;
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < NT) {
        float s = 0.f, bad = 0.f;
        for (int b = 0; b < B; ++b) { s += per_cloud[b * (NT + 1) + threadIdx.x]; bad += per_cloud[b * (NT + 1) + NT]; }
        // a NaN in any fitted plane turns the five box terms into NaN (recon_loss.py:632-639)
        if (bad > 0.f && threadIdx.x >= T_VOTE && threadIdx.x <= T_BB_SELF) s = __builtin_nanf("");
        terms[threadIdx.x] = s;
    }
}

// ---- kernel 4: backward, per-cloud program with one tangent direction per lane -----------------------------------------
// directions: 0..53 moment sums, 54..56 green axis, 57..59 red axis, 60..62 translation, 63..65 size, 66 / 67 confidences
static_assert(NMOM + 14 <= 128, "one lane per tangent direction");
__global__ __launch_bounds__(128) void loss_cloud_bwd_kernel(const float* gt_R, const float* gt_t, const float* gt_s,
                                                             const float* mean_shape, const float* sym, const float* obj_id,
                                                             const float* p_green, const float* p_red, const float* f_green,
                                                             const float* f_red, const float* pred_T, const float* pred_s,
                                                             const float* red, const double* mom, const float* gw,
                                                             HspLossCfg cfg, int B, int N, float* d_mom /* (B,54) */,
                                                             float* d_green, float* d_red, float* d_fg, float* d_fr,
                                                             float* d_T, float* d_s) {
    const int b = blockIdx.x, l = threadIdx.x;
    const float scale2 = rescale_of(sym, B);
    CloudGT c;
    load_gt(c, gt_R, gt_t, gt_s, mean_shape, sym, obj_id, b);
    auto seed = [&](float v, int dir) { return Dual{v, l == dir ? 1.f : 0.f}; };
    Dual g[3], r[3], T[3], s[3], S[NMOM], term[NT];
    for (int i = 0; i < NMOM; ++i) S[i] = seed((float)mom[b * NMOM + i], i);
    for (int i = 0; i < 3; ++i) {
        g[i] = seed(p_green[b * 3 + i], 54 + i);
        r[i] = seed(p_red[b * 3 + i], 57 + i);
        T[i] = seed(pred_T[b * 3 + i], 60 + i);
        s[i] = seed(pred_s[b * 3 + i], 63 + i);
    }
    // the confidences are variables only in R_con (HSPose.py:84-160 detaches them everywhere else): the frames take val()
    const Dual fg = seed(f_green[b], 66), fr = seed(f_red[b], 67);
    float bad;
    cloud_program<Dual>(c, cfg, B, N, scale2, red + b * NRED, S, g, r, fg, fr, T, s, term, &bad);
    float grad = 0.f;
    for (int k = 0; k < NT; ++k) grad += gw[k] * term[k].d;
    if (l < NMOM) d_mom[b * NMOM + l] = grad;
    else if (l < 57) d_green[b * 3 + (l - 54)] = grad;
    else if (l < 60) d_red[b * 3 + (l - 57)] = grad;
    else if (l < 63) d_T[b * 3 + (l - 60)] = grad;
    else if (l < 66) d_s[b * 3 + (l - 63)] = grad;
    else if (l == 66) d_fg[b] = grad;
    else if (l == 67) d_fr[b] = grad;
}

// ---- kernel 5: points pass 2 -- per-point gradients ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_points_bwd_kernel(
    const float* __restrict__ PC, const float* gt_R, const float* gt_t, const float* gt_s, const float* mean_shape,
    const float* sym, const float* obj_id, const float* __restrict__ recon, const float* __restrict__ face_normal,
    const float* __restrict__ face_dis, const float* __restrict__ face_f, const float* p_green, const float* pred_T,
    const float* __restrict__ prm, const float* __restrict__ d_mom, const float* gw, HspLossCfg cfg, int B, int N,
    float* __restrict__ d_recon, float* __restrict__ d_fn, float* __restrict__ d_fd, float* __restrict__ d_ff) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    CloudGT c;
    load_gt(c, gt_R, gt_t, gt_s, mean_shape, sym, obj_id, b);
    float g[3], T[3], nm[3], half[3];
    for (int i = 0; i < 3; ++i) {
        g[i] = p_green[b * 3 + i]; T[i] = pred_T[b * 3 + i]; nm[i] = prm[b * NPRM + P_NM + i];
        half[i] = (c.s[i] + c.ms[i]) / 2.0f;
    }
    const size_t pn = (size_t)b * N + n;
    PointCtx q;
    point_ctx(q, c, T, PC + pn * 3);
    const float fB = (float)B, fN = (float)N;
    {
        float target[3], mir[3], u;
        sym_targets(q, c, g, T, nm, target, mir, u);
        const float crec = gw[T_SYM_RECON] * cfg.prop_sym_w / (3.f * fB * fN), crt = gw[T_SYM_RT] * cfg.prop_sym_w / (3.f * fB * fN);
        for (int i = 0; i < 3; ++i) {
            const float re = recon[pn * 3 + i];
            float gr = 0.f;
            if (!c.skip) gr -= crec * sgnf(target[i] - re);
            if (c.cls_y || c.cls_yx) gr -= crt * sgnf(mir[i] - re);
            d_recon[pn * 3 + i] = gr;
        }
    }
    const float k6 = 1.f / (6.f * fB * fN);
    for (int j = 0; j < 6; ++j) {
        const int a = j % 3, nj = kFacePerm[j];
        const float sign = j < 3 ? 1.f : -1.f;
        const float* fn = face_normal + (pn * 6 + nj) * 3;
        const float fd = face_dis[pn * 6 + nj], ff = face_f[pn * 6 + nj];
        const float ng[3] = {sign * c.R[a], sign * c.R[3 + a], sign * c.R[6 + a]};
        const float dg = half[a] - sign * q.canon[a];
        const float wn = (a == 1 || c.keep) ? gw[T_PER_P] * cfg.recon_n_w * k6 : 0.f;
        const float wd = c.axis_mask[a] ? gw[T_PER_P] * cfg.recon_d_w * k6 : 0.f;
        const float wc = c.axis_mask[a] ? gw[T_P_F] * cfg.recon_f_w * k6 : 0.f;
        float v[3];
        for (int i = 0; i < 3; ++i) v[i] = fn[i] * fd - ng[i] * dg;
        const float err = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const float conf = expf(-303.5f * err * err);
        const float kc = wc * sgnf(conf - ff) * (-607.0f * conf);          // d|conf - ff| = sgn * conf * (-303.5 * 2) (v . dv)
        // votes: moments -> coordinates
        const float* G = d_mom + (size_t)b * NMOM + j * 9;
        const float w = ff, x = q.P[0] + fd * fn[0], y = q.P[1] + fd * fn[1], z = q.P[2] + fd * fn[2];
        const float dv[3] = {w * (2.f * x * G[0] + y * G[1] + G[2] + z * G[6]), w * (x * G[1] + 2.f * y * G[3] + G[4] + z * G[7]),
                             w * (x * G[6] + y * G[7] + G[8])};
        float gfd = wd * sgnf(fd - dg);
        for (int i = 0; i < 3; ++i) {
            d_fn[(pn * 6 + nj) * 3 + i] = -wn * ng[i] + kc * fd * v[i] + dv[i] * fd;
            gfd += kc * v[i] * fn[i] + dv[i] * fn[i];
        }
        d_fd[pn * 6 + nj] = gfd;
        d_ff[pn * 6 + nj] = -wc * sgnf(conf - ff);
    }
}

// ---- on-device augmentation of a training batch (HSPose.data_augment, network/HSPose.py:185-256 over
// datasets/data_augmentation.py:70-190) in one launch: box scaling in the object frame, rigid perturbation, box-cage taper
// (bowls / mugs), per-point radial jitter, each applied to the clouds whose uniform draw falls under its probability.
// One workgroup per cloud; a thread carries a point through the four stages (no cross-point dependency), the tapered
// model's extent (the new size) is a min / max over the model points through LDS.  draws (6,B): u_bb, u_rt, u_bc,
// ey_up, ey_down (raw uniforms, mapped to [0.8, 1.2)), u_pc -- drawn by the caller in the reference's order.
__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ PC, const float* gt_R, const float* gt_t,
                                                      const float* gt_s, const float* mean_shape, const float* sym,
                                                      const float* aug_bb, const float* aug_rt_t, const float* aug_rt_r,
                                                      const float* __restrict__ model_point, const float* nocs_scale,
                                                      const float* obj_id, const float* draws, const float* __restrict__ noise,
                                                      int B, int N, int M, float p_bb, float p_rt, float p_bc, float p_pc,
                                                      float* __restrict__ PC_out, float* R_out, float* t_out, float* s_out) {
    __shared__ float red[6][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float R[9], t[3], s[3], ms[3], ar[9], at[3], k[3];
    for (int i = 0; i < 9; ++i) { R[i] = gt_R[b * 9 + i]; ar[i] = aug_rt_r[b * 9 + i]; }
    for (int i = 0; i < 3; ++i) { t[i] = gt_t[b * 3 + i]; s[i] = gt_s[b * 3 + i]; ms[i] = mean_shape[b * 3 + i]; at[i] = aug_rt_t[b * 3 + i]; }
    const bool f_bb = draws[0 * B + b] < p_bb, f_rt = draws[1 * B + b] < p_rt;
    const float obj = obj_id[b];
    const bool f_bc = draws[2 * B + b] < p_bc && (obj == 5.f || obj == 1.f);
    const float ey_up = draws[3 * B + b] * (1.2f - 0.8f) + 0.8f, ey_down = draws[4 * B + b] * (1.2f - 0.8f) + 0.8f;
    const bool f_pc = draws[5 * B + b] < p_pc;
    // 1. box scaling (data_augmentation.py:70-79): x and z share the mean factor under rotational symmetry
    {
        const float a0 = aug_bb[b * 3], a1 = aug_bb[b * 3 + 1], a2 = aug_bb[b * 3 + 2];
        const bool rs = sym[b * 4] == 1.f;
        k[0] = rs ? (a0 + a2) / 2.0f : a0; k[1] = rs ? (a1 + a1) / 2.0f : a1; k[2] = rs ? (a2 + a0) / 2.0f : a2;
    }
    const float R0[9] = {R[0], R[1], R[2], R[3], R[4], R[5], R[6], R[7], R[8]};
    const float t0[3] = {t[0], t[1], t[2]};
    if (f_bb) for (int i = 0; i < 3; ++i) s[i] = (s[i] + ms[i]) * k[i] - ms[i];
    // 2. rigid perturbation (data_augmentation.py:183-190)
    if (f_rt) {
        float Rn[9], tn[3];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) Rn[3 * i + j] = ar[3 * i] * R[j] + ar[3 * i + 1] * R[3 + j] + ar[3 * i + 2] * R[6 + j];
            tn[i] = ar[3 * i] * (t[0] + at[0]) + ar[3 * i + 1] * (t[1] + at[1]) + ar[3 * i + 2] * (t[2] + at[2]);
        }
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        for (int i = 0; i < 3; ++i) t[i] = tn[i];
    }
    // 3. box-cage taper (data_augmentation.py:108-129): the new size is the extent of the tapered (scaled) model
    const float s_y = s[1] + ms[1];
    auto taper_f = [&](float y) { return (y + s_y / 2.0f) / s_y * (ey_up - ey_down) + ey_down; };
    if (f_bc) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int m = tid; m < M; m += 256) {
            const float* mp = model_point + ((size_t)b * M + m) * 3;
            float v[3] = {mp[0], mp[1], mp[2]};
            if (f_bb) for (int i = 0; i < 3; ++i) v[i] *= k[i];
            const float f = taper_f(v[1]);
            v[0] *= f; v[2] *= f;
            for (int i = 0; i < 3; ++i) { mn[i] = fminf(mn[i], v[i]); mx[i] = fmaxf(mx[i], v[i]); }
        }
        for (int i = 0; i < 3; ++i) { red[i][tid] = mn[i]; red[3 + i][tid] = mx[i]; }
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o)
                for (int i = 0; i < 3; ++i) {
                    red[i][tid] = fminf(red[i][tid], red[i][tid + o]);
                    red[3 + i][tid] = fmaxf(red[3 + i][tid], red[3 + i][tid + o]);
                }
            __syncthreads();
        }
        const float ns = nocs_scale[b];
        for (int i = 0; i < 3; ++i) s[i] = (red[3 + i][0] - red[i][0]) * ns - ms[i];
    }
    if (tid == 0) {
        for (int i = 0; i < 9; ++i) R_out[b * 9 + i] = R[i];
        for (int i = 0; i < 3; ++i) { t_out[b * 3 + i] = t[i]; s_out[b * 3 + i] = s[i]; }
    }
    for (int n = tid; n < N; n += 256) {
        const size_t pn = (size_t)b * N + n;
        float p[3] = {PC[pn * 3], PC[pn * 3 + 1], PC[pn * 3 + 2]};
        if (f_bb) {
            const float d0 = p[0] - t0[0], d1 = p[1] - t0[1], d2 = p[2] - t0[2];
            float o[3];
            for (int j = 0; j < 3; ++j) o[j] = (d0 * R0[j] + d1 * R0[3 + j] + d2 * R0[6 + j]) * k[j];
            for (int i = 0; i < 3; ++i) p[i] = (o[0] * R0[3 * i] + o[1] * R0[3 * i + 1] + o[2] * R0[3 * i + 2]) + t0[i];
        }
        if (f_rt) {
            const float q0 = p[0] + at[0], q1 = p[1] + at[1], q2 = p[2] + at[2];
            for (int i = 0; i < 3; ++i) p[i] = q0 * ar[3 * i] + q1 * ar[3 * i + 1] + q2 * ar[3 * i + 2];
        }
        if (f_bc) {
            const float d0 = p[0] - t[0], d1 = p[1] - t[1], d2 = p[2] - t[2];
            float o[3];
            for (int j = 0; j < 3; ++j) o[j] = d0 * R[j] + d1 * R[3 + j] + d2 * R[6 + j];
            const float f = taper_f(o[1]);
            o[0] *= f; o[2] *= f;
            for (int i = 0; i < 3; ++i) p[i] = (o[0] * R[3 * i] + o[1] * R[3 * i + 1] + o[2] * R[3 * i + 2]) + t[i];
        }
        if (f_pc)
            for (int i = 0; i < 3; ++i) p[i] = p[i] + noise[pn * 3 + i] * (p[i] - t[i]);
        for (int i = 0; i < 3; ++i) PC_out[pn * 3 + i] = p[i];
    }
}

struct LossWs {
    float *prm, *red, *per_cloud;
    double* mom;
};
size_t loss_ws_layout(int B, char* base, LossWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return base ? base + o : nullptr; };
    double* mom = reinterpret_cast<double*>(take((size_t)B * NMOM * sizeof(double)));
    float* prm = reinterpret_cast<float*>(take((size_t)B * NPRM * sizeof(float)));
    float* red = reinterpret_cast<float*>(take((size_t)B * NRED * sizeof(float)));
    float* pc = reinterpret_cast<float*>(take((size_t)B * (NT + 1) * sizeof(float)));
    if (w) { w->mom = mom; w->prm = prm; w->red = red; w->per_cloud = pc; }
    return off;
}

}  // namespace
}  // namespace hsp

using namespace hsp;

extern "C" size_t hsp_pose_losses_workspace_bytes(int B) { return B > 0 ? loss_ws_layout(B, nullptr, nullptr) : 0; }

extern "C" int hsp_pose_losses_fwd(const float* PC, const float* gt_R, const float* gt_t, const float* gt_s,
                                   const float* mean_shape, const float* sym, const float* obj_id, const float* recon,
                                   const float* face_normal, const float* face_dis, const float* face_f,
                                   const float* p_green, const float* p_red, const float* f_green, const float* f_red,
                                   const float* pred_T, const float* pred_s, int B, int N, const HspLossCfg* cfg,
                                   float* terms, void* ws, size_t ws_bytes, hspStream_t stream) {
    if (!PC || !gt_R || !gt_t || !gt_s || !mean_shape || !sym || !obj_id || !recon || !face_normal || !face_dis || !face_f ||
        !p_green || !p_red || !f_green || !f_red || !pred_T || !pred_s || !cfg || !terms || B <= 0 || N <= 0)
        return HSP_ERR_BAD_ARG;
    if (!ws || ws_bytes < loss_ws_layout(B, nullptr, nullptr)) return HSP_ERR_WORKSPACE;
    LossWs w;
    loss_ws_layout(B, reinterpret_cast<char*>(ws), &w);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(loss_prep_kernel, dim3((B + 63) / 64), dim3(64), 0, st, gt_R, gt_t, gt_s, mean_shape, sym, obj_id,
                       p_green, p_red, f_green, f_red, B, w.prm);
    hipLaunchKernelGGL(loss_points_kernel, dim3(B), dim3(PTS_THREADS), 0, st, PC, gt_R, gt_t, gt_s, mean_shape, sym, obj_id,
                       recon, face_normal, face_dis, face_f, p_green, p_red, pred_T, w.prm, N, w.red, w.mom);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, gt_R, gt_t, gt_s, mean_shape, sym, obj_id, p_green, p_red,
                       f_green, f_red, pred_T, pred_s, w.red, w.mom, *cfg, B, N, w.per_cloud, terms);
    return check_launch();
}

extern "C" int hsp_pose_losses_bwd(const float* PC, const float* gt_R, const float* gt_t, const float* gt_s,
                                   const float* mean_shape, const float* sym, const float* obj_id, const float* recon,
                                   const float* face_normal, const float* face_dis, const float* face_f,
                                   const float* p_green, const float* p_red, const float* f_green, const float* f_red,
                                   const float* pred_T, const float* pred_s, int B, int N, const HspLossCfg* cfg,
                                   const float* grad_terms, const void* ws, size_t ws_bytes, float* d_mom_scratch,
                                   float* d_recon, float* d_face_normal, float* d_face_dis, float* d_face_f, float* d_green,
                                   float* d_red, float* d_f_green, float* d_f_red, float* d_T, float* d_s,
                                   hspStream_t stream) {
    if (!PC || !gt_R || !gt_t || !gt_s || !mean_shape || !sym || !obj_id || !recon || !face_normal || !face_dis || !face_f ||
        !p_green || !p_red || !f_green || !f_red || !pred_T || !pred_s || !cfg || !grad_terms || !d_mom_scratch || !d_recon ||
        !d_face_normal || !d_face_dis || !d_face_f || !d_green || !d_red || !d_f_green || !d_f_red || !d_T || !d_s || B <= 0 ||
        N <= 0)
        return HSP_ERR_BAD_ARG;
    if (!ws || ws_bytes < loss_ws_layout(B, nullptr, nullptr)) return HSP_ERR_WORKSPACE;
    LossWs w;
    loss_ws_layout(B, reinterpret_cast<char*>(const_cast<void*>(ws)), &w);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(loss_cloud_bwd_kernel, dim3(B), dim3(128), 0, st, gt_R, gt_t, gt_s, mean_shape, sym, obj_id, p_green,
                       p_red, f_green, f_red, pred_T, pred_s, w.red, w.mom, grad_terms, *cfg, B, N, d_mom_scratch, d_green, d_red,
                       d_f_green, d_f_red, d_T, d_s);
    hipLaunchKernelGGL(loss_points_bwd_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, PC, gt_R, gt_t, gt_s, mean_shape, sym,
                       obj_id, recon, face_normal, face_dis, face_f, p_green, pred_T, w.prm, d_mom_scratch, grad_terms, *cfg, B, N,
                       d_recon, d_face_normal, d_face_dis, d_face_f);
    return check_launch();
}

extern "C" int hsp_pose_augment(const float* PC, const float* gt_R, const float* gt_t, const float* gt_s, const float* mean_shape,
                                const float* sym, const float* aug_bb, const float* aug_rt_t, const float* aug_rt_r,
                                const float* model_point, const float* nocs_scale, const float* obj_id, const float* draws,
                                const float* noise, int B, int N, int M, float p_bb, float p_rt, float p_bc, float p_pc,
                                float* PC_out, float* R_out, float* t_out, float* s_out, hspStream_t stream) {
    if (!PC || !gt_R || !gt_t || !gt_s || !mean_shape || !sym || !aug_bb || !aug_rt_t || !aug_rt_r || !model_point ||
        !nocs_scale || !obj_id || !draws || !noise || !PC_out || !R_out || !t_out || !s_out || B <= 0 || N <= 0 || M <= 0)
        return HSP_ERR_BAD_ARG;
    hipLaunchKernelGGL(augment_kernel, dim3(B), dim3(256), 0, as_stream(stream), PC, gt_R, gt_t, gt_s, mean_shape, sym, aug_bb,
                       aug_rt_t, aug_rt_r, model_point, nocs_scale, obj_id, draws, noise, B, N, M, p_bb, p_rt, p_bc, p_pc, PC_out,
                       R_out, t_out, s_out);
    return check_launch();
}

// ---- the face head's output split (PoseNet9D.py:31-35) ------------------------------------------------------------------------
//   face (R, 30) -> normals (R, 6, 3) = face[:, :18] / ||.||_3 per face,  dis (R, 6) = face[:, 18:24],  conf (R, 6) = sigmoid(face[:, 24:])
// One thread per (row, face): the reference's three slices, a norm, a division and a sigmoid are ~8 element-wise launches forward
// and ~20 in autograd's backward (three zero-padded slice gradients, two adds, the norm's chain) over (R, 30) tensors.
// No epsilon under the norm (a zero normal gives inf / nan as in the reference).
namespace hsp {
__global__ __launch_bounds__(256) void face_split_fwd_kernel(const float* __restrict__ face, long long total, float* __restrict__ nrm,
                                                             float* __restrict__ dis, float* __restrict__ conf) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / 6;
        const int f = (int)(e - r * 6);
        const float* p = face + r * 30;
        const float x = p[3 * f], y = p[3 * f + 1], z = p[3 * f + 2];
        const float n = __fsqrt_rn(x * x + y * y + z * z);
        nrm[e * 3] = x / n; nrm[e * 3 + 1] = y / n; nrm[e * 3 + 2] = z / n;
        dis[e] = p[18 + f];
        conf[e] = 1.0f / (1.0f + __expf(-p[24 + f]));
    }
}
// g_face (R, 30) from the three incoming gradients (any may be null = zero): every element written once
__global__ __launch_bounds__(256) void face_split_bwd_kernel(const float* __restrict__ face, const float* __restrict__ g_nrm,
                                                             const float* __restrict__ g_dis, const float* __restrict__ g_conf,
                                                             long long total, float* __restrict__ g_face) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / 6;
        const int f = (int)(e - r * 6);
        const float* p = face + r * 30;
        float* g = g_face + r * 30;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (g_nrm) {
            const float x = p[3 * f], y = p[3 * f + 1], z = p[3 * f + 2];
            const float n = __fsqrt_rn(x * x + y * y + z * z);
            const float ux = x / n, uy = y / n, uz = z / n;
            const float ax = g_nrm[e * 3], ay = g_nrm[e * 3 + 1], az = g_nrm[e * 3 + 2];
            const float d = ax * ux + ay * uy + az * uz;
            gx = (ax - ux * d) / n; gy = (ay - uy * d) / n; gz = (az - uz * d) / n;      // d(v / |v|) = (I - u u^T) / |v|
        }
        g[3 * f] = gx; g[3 * f + 1] = gy; g[3 * f + 2] = gz;
        g[18 + f] = g_dis ? g_dis[e] : 0.f;
        float gc = 0.f;
        if (g_conf) {
            const float sg = 1.0f / (1.0f + __expf(-p[24 + f]));
            gc = g_conf[e] * sg * (1.0f - sg);
        }
        g[24 + f] = gc;
    }
}
}  // namespace hsp

extern "C" int hsp_face_split_fwd(const float* face, long long R, float* normals, float* dis, float* conf, hspStream_t stream) {
    if (!face || !normals || !dis || !conf || R <= 0) return HSP_ERR_BAD_ARG;
    const long long total = R * 6;
    long long g = (total + 255) / 256;
    if (g > HSP_NUM_CU * 8) g = HSP_NUM_CU * 8;
    hipLaunchKernelGGL(hsp::face_split_fwd_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), face, total, normals, dis, conf);
    return check_launch();
}
extern "C" int hsp_face_split_bwd(const float* face, const float* g_normals, const float* g_dis, const float* g_conf, long long R,
                                  float* g_face, hspStream_t stream) {
    if (!face || !g_face || R <= 0) return HSP_ERR_BAD_ARG;
    const long long total = R * 6;
    long long g = (total + 255) / 256;
    if (g > HSP_NUM_CU * 8) g = HSP_NUM_CU * 8;
    hipLaunchKernelGGL(hsp::face_split_bwd_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), face, g_normals, g_dis, g_conf,
                       total, g_face);
    return check_launch();
}

// ---- a rotation head's output split (PoseNet9D.py:40-46): h (B, 4) -> axis (B, 3) = h[:, 1:] / (||h[:, 1:]|| + 1e-6),
// confidence (B,) = sigmoid(h[:, 0]).  One thread per row, one launch each way (the torch composition: 4 launches forward, ~14
// in autograd's backward, on 16 x 4 numbers).
namespace hsp {
__global__ __launch_bounds__(64) void axis_conf_fwd_kernel(const float* __restrict__ h, int B, float* __restrict__ axis,
                                                           float* __restrict__ conf) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float x = h[4 * b + 1], y = h[4 * b + 2], z = h[4 * b + 3];
    const float d = __fsqrt_rn(x * x + y * y + z * z) + 1e-6f;
    axis[3 * b] = x / d; axis[3 * b + 1] = y / d; axis[3 * b + 2] = z / d;
    conf[b] = 1.0f / (1.0f + __expf(-h[4 * b]));
}
__global__ __launch_bounds__(64) void axis_conf_bwd_kernel(const float* __restrict__ h, const float* __restrict__ g_axis,
                                                           const float* __restrict__ g_conf, int B, float* __restrict__ g_h) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float gx = 0.f, gy = 0.f, gz = 0.f, g0 = 0.f;
    if (g_axis) {
        const float x = h[4 * b + 1], y = h[4 * b + 2], z = h[4 * b + 3];
        const float n = __fsqrt_rn(x * x + y * y + z * z), d = n + 1e-6f;
        const float ax = g_axis[3 * b], ay = g_axis[3 * b + 1], az = g_axis[3 * b + 2];
        // y = v / (|v| + eps):  g_v = g_y / d - (g_y . v) v / (|v| d^2); torch's norm backward gives 0 for the second term at v = 0
        const float s = n > 0.f ? (ax * x + ay * y + az * z) / (n * d * d) : 0.f;
        gx = ax / d - s * x; gy = ay / d - s * y; gz = az / d - s * z;
    }
    if (g_conf) {
        const float sg = 1.0f / (1.0f + __expf(-h[4 * b]));
        g0 = g_conf[b] * sg * (1.0f - sg);
    }
    g_h[4 * b] = g0; g_h[4 * b + 1] = gx; g_h[4 * b + 2] = gy; g_h[4 * b + 3] = gz;
}
}  // namespace hsp

extern "C" int hsp_axis_conf_fwd(const float* h, int B, float* axis, float* conf, hspStream_t stream) {
    if (!h || !axis || !conf || B <= 0) return HSP_ERR_BAD_ARG;
    hipLaunchKernelGGL(hsp::axis_conf_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), h, B, axis, conf);
    return check_launch();
}
extern "C" int hsp_axis_conf_bwd(const float* h, const float* g_axis, const float* g_conf, int B, float* g_h, hspStream_t stream) {
    if (!h || !g_h || B <= 0) return HSP_ERR_BAD_ARG;
    hipLaunchKernelGGL(hsp::axis_conf_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), h, g_axis, g_conf, B, g_h);
    return check_launch();
}

