"""The BLAS-library comparison of bench.py and of the tests' second GEMM mode -- NOT part of the product.

hs_pose_amd/ops.py computes every dense product on hand-written kernels.  ``enable()`` swaps the layer composites of that module
for calls into the BLAS library through torch (hipBLASLt / rocBLAS, TunableOp-selected when ``tune``) and REPLACES the ``*_ok``
predicates of the fusions only the hand-written kernels offer (BatchNorm first pass in a product's epilogue, grouped input
gradients, ...) with ``False``: the figure a user would get from the same network on stock PyTorch GEMMs.  ``ops.gemm_mode`` (an
attribute this shim adds; the product has no such notion) says "library" afterwards.
bench.py runs it in a child process (HSP_GEMM=library) and reports ``library_gemm_ms_per_step`` next to the headline."""
import torch


def enable(monkeypatch=None, tune=False):
    from hs_pose_amd import ops

    def put(name, fn):
        if monkeypatch is not None:
            monkeypatch.setattr(ops, name, fn)
        else:
            setattr(ops, name, fn)

    def fm_rows(X2, weights, bias, out=None):
        return torch.addmm(bias, X2, weights) if out is None else torch.addmm(bias, X2, weights, out=out)

    def layer_out_rows_plain(x2, w_ste, F2, Wa, t2, out3, relu=False):
        B, N, C = out3.shape
        out = out3.view(B * N, C)
        torch.mm(x2, w_ste.t(), out=out)
        out.addmm_(F2, Wa.t())
        ops._residual_bias(out3, F2.view(B, N, C), t2)
        return torch.relu_(out3) if relu else out3

    def mm_nn(g2, W, out=None, alpha=1.0):
        if out is None:
            out = torch.empty(g2.shape[0], W.shape[1], dtype=g2.dtype, device=g2.device)
        if alpha == 1.0:
            return torch.mm(g2, W, out=out)
        return torch.addmm(out, g2, W, beta=0.0, alpha=alpha, out=out)

    def mm_nt(x2, W, bias=None, out=None):
        if out is None:
            return torch.addmm(bias, x2, W.t()) if bias is not None else torch.mm(x2, W.t())
        if bias is not None:
            return torch.addmm(bias, x2, W.t(), out=out)
        return torch.mm(x2, W.t(), out=out)

    def grad_in_rows(g2, w_ste, gfm2, weights, out):
        torch.mm(g2, w_ste, out=out)
        return out.addmm_(gfm2, weights.t())

    def tiny_tn(a, b, out, mom=None, gste=None):
        torch.mm(a.t(), b, out=out)
        if mom is not None:
            Cm = mom.shape[1] // 4
            gste.copy_(mom[:, Cm:].sum(dim=0).view(3, Cm).t())
        return out

    own_wgrad = ops.wgrad

    def wgrad(A2, B2, out=None, colsum=False):
        K, M = A2.shape
        if ops._wgrad_ok(A2, B2, out if out is not None else A2):       # the parameter gradients stay on the split-K kernels
            return own_wgrad(A2, B2, out=out, colsum=colsum)             # (as in every round's library-mode figure)
        if out is None:
            out = torch.empty(M, B2.shape[1], dtype=torch.float32, device=A2.device)
        if out.is_contiguous():
            torch.mm(A2.t(), B2, out=out)
        else:
            out.copy_(A2.t() @ B2)
        return (out, B2.sum(dim=0)) if colsum else out

    def never(*a, **kw):
        return False

    for name, fn in (("_fm_rows", fm_rows), ("_layer_out_rows_plain", layer_out_rows_plain), ("_mm_nn", mm_nn), ("_mm_nt", mm_nt),
                     ("_grad_in_rows", grad_in_rows), ("_tiny_tn", tiny_tn), ("wgrad", wgrad), ("gemm_mode", "library"),
                     # the own-kernel fusions: off
                     ("_wgrad_ragged_ok", never), ("linear_bn_part_ok", never), ("_layer_out_bn_ok", never), ("_ste_moments_ok", never),
                     ("_thin_wgrad_ok", never), ("fan_linear_rows_ok", never), ("cloud_cat_linear_ok", never),
                     ("x3_refresh", lambda: None)):
        if monkeypatch is not None and not hasattr(ops, name):
            monkeypatch.setattr(ops, name, fn, raising=False)
        else:
            put(name, fn)
    if tune:
        from tools import gemm_tuning
        gemm_tuning.enable()
