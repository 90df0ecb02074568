// hsp_torch.cpp -- the thin PyTorch-ROCm binding over libhsp.so's C-ABI (include/hsp.h), SURVEY.md section 8(b).
//
// Modelled on the reference's only FFI, tools/pyTorchChamferDistance/chamfer_distance.cpp:180-185 (a pybind module of free functions
// over at::Tensor): each function here does ONLY what that file does for its four entry points, plus what it forgot --
// TORCH_CHECK of device / dtype / contiguity / shapes, output allocation, the CURRENT HIP stream (the reference used stream 0
// implicitly), int32 indices at the C boundary, and a non-zero return code raised as a RuntimeError (the reference printf'd CUDA
// errors, chamfer_distance.cu:152-154).  No arithmetic lives in this file; every launch is an hsp_* symbol.
//
//   * forward / forward_cuda / backward / backward_cuda: the reference module's own names and argument roles (caller-allocated
//     outputs filled in place) -- `import _hsp_torch as cd` is a drop-in for the extension chamfer_distance.py:8-10 JIT-builds.
//   * get_neighbor_index / get_nearest_index: gcn3d.py:15-36 (int64 out, like the reference's API).
//   * the eval-mode HS layers as ONE call each (hs_layer_forward, surface_layer_forward, pool_forward, bn_eval, center_cloud): the
//     launch sequence of hs_pose_amd/ops.py::_HSLayer.forward in the exact (reference-order) form, issued from C++ -- the host side
//     of an inference forward drops from ~10 Python-level launches per layer to one (eager inference is host-bound).
// The ctypes binding (hs_pose_amd/_lib.py) stays: it is what the ABI tests and the training path call.
#include <torch/extension.h>
#include <ATen/hip/HIPContext.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include "hsp.h"

namespace {

hspStream_t cur_stream() { return reinterpret_cast<hspStream_t>(c10::hip::getCurrentHIPStream().stream()); }

void ok(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, ": ", hsp_error_string(rc), " (", hsp_last_hip_error(), ")");
}

void want(const at::Tensor& t, const char* name, at::ScalarType dt, int64_t dims = -1) {
    TORCH_CHECK(t.defined(), name, ": undefined tensor");
    TORCH_CHECK(t.is_cuda(), name, ": expected a GPU tensor (there is no CPU path)");
    TORCH_CHECK(t.scalar_type() == dt, name, ": expected dtype ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, ": expected a contiguous tensor");
    TORCH_CHECK(dims < 0 || t.dim() == dims, name, ": expected ", dims, " dimensions, got ", t.dim());
}

// every entry point: the device of its first tensor becomes current for the call (streams, allocations and launches then agree
// with the tensors even when the caller's current device is another GPU), and every other tensor must live there too
struct OnDevice {
    c10::OptionalDeviceGuard guard;            // (the generic guard: ROCm torch registers its GPU guard under the "cuda" device type)
    c10::Device dev;
    explicit OnDevice(const at::Tensor& first, const char* name) : dev(c10::kCPU) {
        TORCH_CHECK(first.defined() && first.is_cuda(), name, ": expected a GPU tensor (there is no CPU path)");
        dev = first.device();
        guard.reset_device(dev);
    }
    void same(std::initializer_list<std::pair<const at::Tensor*, const char*>> ts) const {
        for (const auto& t : ts)
            TORCH_CHECK(t.first->defined() && t.first->device() == dev, t.second, ": expected a tensor on ", dev, ", got ",
                        t.first->defined() ? t.first->device().str() : std::string("an undefined tensor"));
    }
};
#define HSP_T(x) std::make_pair(&(x), #x)

// rows of a 2-D fp32 matrix that may be a column block of a wider one (leading dimension = stride(0))
int ld_of(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.dim() == 2 && t.stride(1) == 1, name, ": expected a 2-D matrix with contiguous rows");
    return (int)(t.size(0) > 1 ? t.stride(0) : std::max<int64_t>(t.stride(0), t.size(1)));
}
const float* fp(const at::Tensor& t) { return t.data_ptr<float>(); }

at::Tensor bytes_ws(size_t n, const at::Tensor& like) {
    return at::empty({(int64_t)std::max<size_t>(n, 16)}, like.options().dtype(at::kByte));
}

// ---------------------------------------------------------------------------------------------------------------------
// the reference's pybind surface (chamfer_distance.cpp:27-56, :90-177): xyz1 (B,n,3), xyz2 (B,m,3), outputs filled in place
// ---------------------------------------------------------------------------------------------------------------------
void chamfer_forward(at::Tensor xyz1, at::Tensor xyz2, at::Tensor dist1, at::Tensor dist2, at::Tensor idx1, at::Tensor idx2) {
    const OnDevice on(xyz1, "xyz1");
    on.same({HSP_T(xyz2), HSP_T(dist1), HSP_T(dist2), HSP_T(idx1), HSP_T(idx2)});
    want(xyz1, "xyz1", at::kFloat, 3); want(xyz2, "xyz2", at::kFloat, 3);
    want(dist1, "dist1", at::kFloat, 2); want(dist2, "dist2", at::kFloat, 2);
    want(idx1, "idx1", at::kInt, 2); want(idx2, "idx2", at::kInt, 2);
    const int B = (int)xyz1.size(0), n = (int)xyz1.size(1), m = (int)xyz2.size(1);
    TORCH_CHECK(xyz1.size(2) == 3 && xyz2.size(2) == 3 && xyz2.size(0) == B, "chamfer: expected (B,n,3) and (B,m,3)");
    TORCH_CHECK(dist1.size(0) == B && dist1.size(1) == n && idx1.sizes() == dist1.sizes(), "chamfer: dist1 / idx1 must be (B,n)");
    TORCH_CHECK(dist2.size(0) == B && dist2.size(1) == m && idx2.sizes() == dist2.sizes(), "chamfer: dist2 / idx2 must be (B,m)");
    ok(hsp_chamfer_fwd(fp(xyz1), fp(xyz2), B, n, m, dist1.data_ptr<float>(), dist2.data_ptr<float>(), idx1.data_ptr<int32_t>(),
                       idx2.data_ptr<int32_t>(), cur_stream()), "hsp_chamfer_fwd");
}
void chamfer_backward(at::Tensor xyz1, at::Tensor xyz2, at::Tensor gradxyz1, at::Tensor gradxyz2, at::Tensor graddist1,
                      at::Tensor graddist2, at::Tensor idx1, at::Tensor idx2) {
    const OnDevice on(xyz1, "xyz1");
    on.same({HSP_T(xyz2), HSP_T(gradxyz1), HSP_T(gradxyz2), HSP_T(graddist1), HSP_T(graddist2), HSP_T(idx1), HSP_T(idx2)});
    want(xyz1, "xyz1", at::kFloat, 3); want(xyz2, "xyz2", at::kFloat, 3);
    want(gradxyz1, "gradxyz1", at::kFloat, 3); want(gradxyz2, "gradxyz2", at::kFloat, 3);
    want(graddist1, "graddist1", at::kFloat, 2); want(graddist2, "graddist2", at::kFloat, 2);
    want(idx1, "idx1", at::kInt, 2); want(idx2, "idx2", at::kInt, 2);
    const int B = (int)xyz1.size(0), n = (int)xyz1.size(1), m = (int)xyz2.size(1);
    TORCH_CHECK(xyz1.size(2) == 3 && xyz2.size(2) == 3 && xyz2.size(0) == B, "chamfer: expected (B,n,3) and (B,m,3)");
    TORCH_CHECK(gradxyz1.sizes() == xyz1.sizes() && gradxyz2.sizes() == xyz2.sizes(), "chamfer: gradient shapes");
    TORCH_CHECK(graddist1.size(0) == B && graddist1.size(1) == n && idx1.sizes() == graddist1.sizes(), "chamfer: graddist1 / idx1 must be (B,n)");
    TORCH_CHECK(graddist2.size(0) == B && graddist2.size(1) == m && idx2.sizes() == graddist2.sizes(), "chamfer: graddist2 / idx2 must be (B,m)");
    ok(hsp_chamfer_bwd(fp(xyz1), fp(xyz2), idx1.data_ptr<int32_t>(), idx2.data_ptr<int32_t>(), fp(graddist1), fp(graddist2), B, n, m,
                       gradxyz1.data_ptr<float>(), gradxyz2.data_ptr<float>(), cur_stream()), "hsp_chamfer_bwd");
}

// ---------------------------------------------------------------------------------------------------------------------
// gcn3d.py:15-36
// ---------------------------------------------------------------------------------------------------------------------
at::Tensor knn_i32(const at::Tensor& x, int k, bool drop_first, bool exact, bool transposed_view) {
    const OnDevice on(x, "vertices");
    want(x, "vertices", at::kFloat, 3);
    const int B = (int)x.size(0), N = (int)x.size(1), C = (int)x.size(2);
    TORCH_CHECK(k >= 1 && k + (drop_first ? 1 : 0) <= N, "get_neighbor_index: k out of range");
    auto idx = at::empty({B, N, k}, x.options().dtype(at::kInt));
    if (exact && k + (drop_first ? 1 : 0) + 1 <= 33) {
        const size_t wsb = hsp_knn_exact_workspace_bytes(B, N, C, k, drop_first);
        auto ws = bytes_ws(wsb, x);
        ok(hsp_knn_exact_f32(fp(x), B, N, C, k, drop_first, transposed_view ? 1 : 0, idx.data_ptr<int32_t>(), ws.data_ptr(), wsb,
                             nullptr, cur_stream()), "hsp_knn_exact_f32");
    } else if (C == 3 && k + (drop_first ? 1 : 0) + 1 <= 33 && N >= 2 && (size_t)N * 16 <= 160 * 1024) {
        // coordinates: torch.topk's order among equal distances, like every xyz search of the package (ops.knn_xyz)
        const size_t wsb = hsp_knn_xyz_workspace_bytes(B, N);
        auto ws = bytes_ws(wsb, x);
        ok(hsp_knn_xyz_f32(fp(x), B, N, k, 0, drop_first, idx.data_ptr<int32_t>(), nullptr, ws.data_ptr(), wsb, nullptr, cur_stream()),
           "hsp_knn_xyz_f32");
    } else {
        const size_t wsb = hsp_knn_workspace_bytes(B, N, C, k);
        auto ws = bytes_ws(wsb, x);
        ok(hsp_knn_f32(fp(x), B, N, C, k, drop_first, idx.data_ptr<int32_t>(), ws.data_ptr(), wsb, cur_stream()), "hsp_knn_f32");
    }
    return idx;
}
at::Tensor get_neighbor_index(at::Tensor vertices, int64_t neighbor_num) {      // int64, as the reference returns it
    return knn_i32(vertices, (int)neighbor_num, true, false, false).to(at::kLong);
}
at::Tensor get_nearest_index(at::Tensor target, at::Tensor source) {
    const OnDevice on(target, "target");
    on.same({HSP_T(source)});
    want(target, "target", at::kFloat, 3); want(source, "source", at::kFloat, 3);
    TORCH_CHECK(target.size(2) == 3 && source.size(2) == 3 && source.size(0) == target.size(0), "get_nearest_index: (B,Nt,3), (B,Ns,3)");
    auto idx = at::empty({target.size(0), target.size(1)}, target.options().dtype(at::kInt));
    ok(hsp_nn1_f32(fp(target), (int)target.size(1), fp(source), (int)source.size(1), (int)target.size(0), idx.data_ptr<int32_t>(),
                   cur_stream()), "hsp_nn1_f32");
    return idx.to(at::kLong).unsqueeze(-1);
}

// ---------------------------------------------------------------------------------------------------------------------
// eval-mode forward pieces (no autograd: inference), exact (reference-order) form -- see hs_pose_amd/ops.py::exact_scope
// ---------------------------------------------------------------------------------------------------------------------
at::Tensor gemm_wave(const at::Tensor& A, const at::Tensor& Bm, bool nn, const at::Tensor* bias) {
    const int M = (int)A.size(0), K = (int)A.size(1), N = (int)(nn ? Bm.size(1) : Bm.size(0));
    auto out = at::empty({M, N}, A.options());
    ok(hsp_gemm_wave_f32(fp(A), ld_of(A, "A"), fp(Bm), ld_of(Bm, "B"), nn ? 1 : 0, K, nullptr, 0, nullptr, 0, 0, 0, M, N,
                         bias ? fp(*bias) : nullptr, nullptr, 0, nullptr, 0, 1.0f, nullptr, nullptr, out.data_ptr<float>(), N, 0,
                         cur_stream()), "hsp_gemm_wave_f32");
    return out;
}

std::tuple<at::Tensor, at::Tensor> orl_exact(const at::Tensor& F3, const at::Tensor& idx_x, int k) {
    const int B = (int)F3.size(0), N = (int)F3.size(1), C = (int)F3.size(2);
    auto fg = at::empty({B, C}, F3.options());
    auto arg = at::empty({B, N, C}, F3.options().dtype(at::kByte));
    const size_t wsb = hsp_orl_exact_workspace_bytes(B, N, C);
    auto ws = bytes_ws(wsb, F3);
    ok(hsp_orl_global_exact_f32(fp(F3), idx_x.data_ptr<int32_t>(), B, N, k, (int)idx_x.size(2), C, fg.data_ptr<float>(),
                                arg.data_ptr<uint8_t>(), ws.data_ptr(), wsb, cur_stream()), "hsp_orl_global_exact_f32");
    return {fg, arg};
}

void layer_out_exact(const at::Tensor& F2, const at::Tensor& w_conv2, const at::Tensor& fg, int N, at::Tensor& out2, const at::Tensor* ste,
                     const at::Tensor* xyz3, const at::Tensor* w3, bool relu) {
    const int R = (int)F2.size(0), C = (int)F2.size(1);
    auto Wa = w_conv2.narrow(1, 0, C), Wb = w_conv2.narrow(1, C, C);
    const bool two = 2 * C > 256;
    at::Tensor t2;
    if (two) t2 = gemm_wave(fg, Wb, false, nullptr);             // K = 512: the f_global block is its own chain
    ok(hsp_layer_out_exact_f32(fp(F2), ld_of(F2, "F"), fp(Wa), ld_of(Wa, "Wa"), two ? nullptr : fp(fg), two ? 0 : ld_of(fg, "fg"),
                               two ? nullptr : fp(Wb), two ? 0 : ld_of(Wb, "Wb"), two ? fp(t2) : nullptr, two ? 1 : 0,
                               ste ? fp(*ste) : nullptr, ste ? ld_of(*ste, "ste") : 0, xyz3 ? fp(*xyz3) : nullptr, w3 ? fp(*w3) : nullptr,
                               relu ? 1 : 0, R, C, N, out2.data_ptr<float>(), C, cur_stream()), "hsp_layer_out_exact_f32");
}

bool exact_shapes(int N, int Cin, int C) { return C % 32 == 0 && (Cin == 3 || Cin % 32 == 0) && 2 * C <= 512 && N >= 32; }

// HS_layer.forward (gcn3d.py:143-156), eval mode.  idx_f: the feature-space neighbour index (exactly k columns; knn_exact),
// idx_x: the xyz one (>= k columns), w_ste (Cout,Cin[,1]), w_conv2 (Cout,2Cout[,1]).  Returns out (B,N,Cout).
at::Tensor hs_layer_forward(at::Tensor xyz, at::Tensor X, at::Tensor idx_f, at::Tensor idx_x, int64_t k, int64_t S, at::Tensor weights,
                            at::Tensor bias, at::Tensor directions, at::Tensor w_ste, at::Tensor w_conv2) {
    const OnDevice on(X, "feature_map");
    on.same({HSP_T(xyz), HSP_T(idx_f), HSP_T(idx_x), HSP_T(weights), HSP_T(bias), HSP_T(directions), HSP_T(w_ste), HSP_T(w_conv2)});
    want(xyz, "vertices", at::kFloat, 3); want(X, "feature_map", at::kFloat, 3); want(idx_f, "idx_f", at::kInt, 3); want(idx_x, "idx_x", at::kInt, 3);
    want(weights, "weights", at::kFloat, 2); want(bias, "bias", at::kFloat, 1); want(directions, "directions", at::kFloat, 2);
    if (w_ste.dim() == 3) w_ste = w_ste.squeeze(-1);
    if (w_conv2.dim() == 3) w_conv2 = w_conv2.squeeze(-1);
    want(w_ste, "STE_layer.weight", at::kFloat, 2); want(w_conv2, "conv2.weight", at::kFloat, 2);
    TORCH_CHECK(S > 0 && directions.size(0) == 3 && directions.size(1) % S == 0, "HS_layer: directions must be (3, S Cout)");
    TORCH_CHECK(xyz.size(0) == X.size(0) && xyz.size(1) == X.size(1) && xyz.size(2) == 3, "HS_layer: vertices must be (B,N,3)");
    const int B = (int)X.size(0), N = (int)X.size(1), Cin = (int)X.size(2), SC = (int)directions.size(1), C = SC / (int)S;
    TORCH_CHECK(weights.size(0) == Cin && weights.size(1) == (S + 1) * C, "HS_layer: weights must be (Cin, (S+1) Cout)");
    TORCH_CHECK(bias.size(0) == (S + 1) * C, "HS_layer: bias must hold (S+1) Cout entries");
    TORCH_CHECK(idx_x.size(0) == B && idx_x.size(1) == N, "HS_layer: idx_x must be (B,N,>=k)");
    TORCH_CHECK(w_ste.size(0) == C && w_ste.size(1) == Cin && w_conv2.size(0) == C && w_conv2.size(1) == 2 * C, "HS_layer: STE / conv2 shapes");
    TORCH_CHECK(exact_shapes(N, Cin, C), "hs_layer_forward: shape outside the reference-order forms (use the Python path)");
    TORCH_CHECK(idx_x.size(2) >= k && idx_f.size(2) == k && idx_f.size(0) == B && idx_f.size(1) == N, "HS_layer: idx_f (B,N,k), idx_x (B,N,>=k)");
    auto X2 = X.view({B * N, Cin});
    auto fm = gemm_wave(X2, weights, true, &bias);                                          // gcn3d.py:171
    auto F3 = at::empty({B, N, C}, X.options());
    auto arg = at::empty({B, N, SC}, X.options().dtype(at::kUInt16));
    ok(hsp_rf_conv_fwd(fp(xyz), idx_f.data_ptr<int32_t>(), fp(directions), fp(fm), B, N, (int)k, (int)S, C, F3.data_ptr<float>(),
                       reinterpret_cast<uint16_t*>(arg.data_ptr()), nullptr, cur_stream()), "hsp_rf_conv_fwd");
    auto fgarg = orl_exact(F3, idx_x, (int)k);
    auto ste = gemm_wave(X2, w_ste, false, nullptr);                                        // gcn3d.py:149
    auto out = at::empty({B, N, C}, X.options());
    auto out2 = out.view({B * N, C});
    layer_out_exact(F3.view({B * N, C}), w_conv2, std::get<0>(fgarg), N, out2, &ste, nullptr, nullptr, false);
    return out;
}

// HSlayer_surface.forward (gcn3d.py:79-90), eval mode; relu: FaceRecon.py:88's relu inside the last product
at::Tensor surface_layer_forward(at::Tensor xyz, at::Tensor idx_x, int64_t k, int64_t S, at::Tensor directions, at::Tensor w_ste,
                                 at::Tensor w_conv2, bool relu) {
    const OnDevice on(xyz, "vertices");
    on.same({HSP_T(idx_x), HSP_T(directions), HSP_T(w_ste), HSP_T(w_conv2)});
    want(xyz, "vertices", at::kFloat, 3); want(idx_x, "idx_x", at::kInt, 3); want(directions, "directions", at::kFloat, 2);
    if (w_ste.dim() == 3) w_ste = w_ste.squeeze(-1);
    if (w_conv2.dim() == 3) w_conv2 = w_conv2.squeeze(-1);
    want(w_ste, "STE_layer.weight", at::kFloat, 2); want(w_conv2, "conv2.weight", at::kFloat, 2);
    TORCH_CHECK(S > 0 && directions.size(0) == 3 && directions.size(1) % S == 0 && xyz.size(2) == 3, "HSlayer_surface: (B,N,3), directions (3, S K)");
    const int B = (int)xyz.size(0), N = (int)xyz.size(1), SC = (int)directions.size(1), C = SC / (int)S;
    TORCH_CHECK(w_ste.size(0) == C && w_ste.size(1) == 3 && w_conv2.size(0) == C && w_conv2.size(1) == 2 * C, "HSlayer_surface: STE / conv2 shapes");
    TORCH_CHECK(idx_x.size(0) == B && idx_x.size(1) == N && idx_x.size(2) == k, "HSlayer_surface: idx must be (B,N,k) with exactly k columns");
    TORCH_CHECK(exact_shapes(N, 3, C), "surface_layer_forward: shape outside the reference-order forms (use the Python path)");
    auto F3 = at::empty({B, N, C}, xyz.options());
    auto arg = at::empty({B, N, SC}, xyz.options().dtype(at::kUInt16));
    ok(hsp_rf_surface_fwd(fp(xyz), idx_x.data_ptr<int32_t>(), fp(directions), B, N, (int)k, (int)S, C, F3.data_ptr<float>(),
                          reinterpret_cast<uint16_t*>(arg.data_ptr()), cur_stream()), "hsp_rf_surface_fwd");
    auto fgarg = orl_exact(F3, idx_x, (int)k);
    auto out = at::empty({B, N, C}, xyz.options());
    auto out2 = out.view({B * N, C});
    auto x2 = xyz.view({B * N, 3});
    auto w3 = w_ste.contiguous();
    layer_out_exact(F3.view({B * N, C}), w_conv2, std::get<0>(fgarg), N, out2, nullptr, &x2, &w3, relu);
    return out;
}

// Pool_layer.forward (gcn3d.py:226-246) for the kept rows sel (int32, drawn by the caller exactly as the reference draws them)
std::tuple<at::Tensor, at::Tensor> pool_forward(at::Tensor xyz, at::Tensor feat, at::Tensor idx_x, at::Tensor sel, int64_t k) {
    const OnDevice on(feat, "feature_map");
    on.same({HSP_T(xyz), HSP_T(idx_x), HSP_T(sel)});
    want(xyz, "vertices", at::kFloat, 3); want(feat, "feature_map", at::kFloat, 3); want(idx_x, "idx_x", at::kInt, 3); want(sel, "sel", at::kInt, 1);
    const int B = (int)feat.size(0), N = (int)feat.size(1), C = (int)feat.size(2), Nq = (int)sel.size(0);
    TORCH_CHECK(xyz.size(0) == B && xyz.size(1) == N && xyz.size(2) == 3, "Pool_layer: vertices must be (B,N,3)");
    TORCH_CHECK(k > 0 && idx_x.size(0) == B && idx_x.size(1) == N && idx_x.size(2) >= k, "Pool_layer: idx must be (B,N,>=k)");
    // (the kept-row ids are device data: hsp_pool_fwd reads sel[q] as a row of the N; a caller-side range check would cost a sync --
    // the Python callers draw them with torch.randperm(N), gcn3d.py:243)
    auto out = at::empty({B, Nq, C}, feat.options());
    auto arg = at::empty({B, Nq, C}, feat.options().dtype(at::kByte));
    auto v = at::empty({B, Nq, 3}, xyz.options());
    ok(hsp_pool_fwd(fp(feat), fp(xyz), idx_x.data_ptr<int32_t>(), sel.data_ptr<int32_t>(), B, N, Nq, (int)k, (int)idx_x.size(2), C,
                    out.data_ptr<float>(), arg.data_ptr<uint8_t>(), v.data_ptr<float>(), cur_stream()), "hsp_pool_fwd");
    return {v, out};
}

// eval-mode BatchNorm1d (+ relu) on point rows; invstd: the host's 1 / sqrt(running_var + eps) (may be undefined)
at::Tensor bn_eval(at::Tensor x, at::Tensor running_mean, at::Tensor running_var, c10::optional<at::Tensor> invstd, at::Tensor weight,
                   at::Tensor bias, double eps, bool relu) {
    const OnDevice on(x, "x");
    on.same({HSP_T(running_mean), HSP_T(running_var), HSP_T(weight), HSP_T(bias)});
    want(x, "x", at::kFloat);
    want(running_mean, "running_mean", at::kFloat, 1); want(running_var, "running_var", at::kFloat, 1);
    want(weight, "weight", at::kFloat, 1); want(bias, "bias", at::kFloat, 1);
    if (invstd.has_value()) {
        on.same({std::make_pair(&*invstd, "invstd")});
        want(*invstd, "invstd", at::kFloat, 1);
    }
    const int C = (int)x.size(-1);
    TORCH_CHECK(!invstd.has_value() || invstd->numel() == C, "bn_eval: invstd must hold one entry per channel");
    TORCH_CHECK(running_mean.numel() == C && running_var.numel() == C && weight.numel() == C && bias.numel() == C, "bn_eval: channel counts");
    auto y = at::empty_like(x);
    ok(hsp_bn_eval_f32(fp(x), x.numel() / C, C, fp(running_mean), fp(running_var), invstd.has_value() ? fp(*invstd) : nullptr, fp(weight),
                       fp(bias), (float)eps, relu ? 1 : 0, y.data_ptr<float>(), cur_stream()), "hsp_bn_eval_f32");
    return y;
}

// PoseNet9D.py:25
std::tuple<at::Tensor, at::Tensor> center_cloud(at::Tensor points) {
    const OnDevice on(points, "points");
    want(points, "points", at::kFloat, 3);
    TORCH_CHECK(points.size(2) == 3, "center_cloud: (B,N,3)");
    auto out = at::empty_like(points);
    auto mean = at::empty({points.size(0), 1, 3}, points.options());
    ok(hsp_center_cloud_f32(fp(points), (int)points.size(0), (int)points.size(1), out.data_ptr<float>(), mean.data_ptr<float>(), cur_stream()),
       "hsp_center_cloud_f32");
    return {out, mean};
}

at::Tensor knn_exact(at::Tensor x, int64_t k, bool drop_first, bool transposed_view) {
    return knn_i32(x, (int)k, drop_first, true, transposed_view);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "thin PyTorch-ROCm binding over libhsp.so (include/hsp.h)";
    // the reference extension's surface (chamfer_distance.cpp:180-185); there is no CPU implementation behind the plain names
    m.def("forward", &chamfer_forward, "ChamferDistance forward (HIP)");
    m.def("forward_cuda", &chamfer_forward, "ChamferDistance forward (HIP)");
    m.def("backward", &chamfer_backward, "ChamferDistance backward (HIP)");
    m.def("backward_cuda", &chamfer_backward, "ChamferDistance backward (HIP)");
    m.def("get_neighbor_index", &get_neighbor_index, "gcn3d.get_neighbor_index -> int64 (B,N,k)");
    m.def("get_nearest_index", &get_nearest_index, "gcn3d.get_nearest_index -> int64 (B,Nt,1)");
    m.def("knn_exact", &knn_exact, "int32 neighbour index with torch.topk's tie order (eval-mode forward)");
    m.def("hs_layer_forward", &hs_layer_forward, "HS_layer.forward, eval mode, reference-order arithmetic");
    m.def("surface_layer_forward", &surface_layer_forward, "HSlayer_surface.forward, eval mode, reference-order arithmetic");
    m.def("pool_forward", &pool_forward, "Pool_layer.forward for given kept rows");
    m.def("bn_eval", &bn_eval, "eval-mode BatchNorm1d (+ relu) on point rows");
    m.def("center_cloud", &center_cloud, "points - mean over the points (reference summation order)");
}
