"""Fused grouping + shared MLP + max-pool of a set-abstraction layer on MI355X.

Computes exactly what `QueryAndGroup` -> `SharedMLP(bn=True)` -> `F.max_pool2d` compute in the
reference (pointnet2_utils.py:317-376, pytorch_utils.py:11-36, pointnet2_modules.py:243-267,
train-mode BatchNorm with batch statistics), forward and backward, but through the `btr_sa_*`
kernels of libbtr_pointnet2.so (csrc/sa_mlp.hip): channel-last activations, f32-MFMA GEMMs
with the previous layer's BN+ReLU fused into the operand load and the BN statistics fused
into the epilogue; the (B, 3+C, npoint, nsample) tensor and the post-activation tensors never
exist in HBM.  Used by `PointnetSAModuleVotes` when its configuration allows (see
`can_fuse`); `BTR_FUSED_SA=0` disables it (the unfused path runs the nine `_ext` ops + torch).
"""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function

if __package__:
    from . import _ext
    from . import grad_sink
    from . import pointnet2_utils
else:  # top-level import, like the reference's scripts (sys.path.append(.../pointnet2))
    import pointnet2._ext as _ext
    import grad_sink
    import pointnet2_utils
_call, _lib, _on, _p, _stream = _ext._call, _ext._lib, _ext._on, _ext._p, _ext._stream


def enabled():
    return os.environ.get("BTR_FUSED_SA", "1") != "0"


def _ceil4(v):
    return (v + 3) // 4 * 4


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


class FusedSAFunction(Function):
    """forward(xyz, new_xyz, features|None, idx, meta, *params) -> (B, C_last, M)

    meta: dict(radius_div, use_xyz, bns=[nn.BatchNorm2d...]); params = (W0, g0, b0, W1, ...)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx, meta, *params):
        _ext.RUNNING_STATS_EPOCH[0] += 1   # running statistics move through raw pointers
        dev = xyz.device
        B, N, _ = xyz.shape
        M, S = idx.shape[1], idx.shape[2]
        C = features.shape[1] if features is not None else 0
        use_xyz = 1 if meta["use_xyz"] else 0
        rdiv = float(meta["radius_div"])
        bns = meta["bns"]
        L = len(params) // 3
        R = B * M * S
        K0 = 3 * use_xyz + C
        K0p = _ceil4(K0)
        xyz = xyz.contiguous()
        new_xyz = new_xyz.contiguous()
        feats_cl = None
        if C:
            feats_cl = _ext.twin_of(features)
            if os.environ.get("BTR_SA_CL_SHORTCUT", "1") == "0":
                feats_cl = None
            if feats_cl is None or feats_cl.shape != (B, N, C):
                feats_cl = features.transpose(1, 2).contiguous()

        # Compact rows (csrc/sa_mlp.hip "compact rows"): layers whose groups are mostly padding
        # (nsample >= 32: SA1 keeps 67 %, SA2 43 % of its rows at the benchmark shape) are
        # evaluated on the distinct neighbours only.  Needs the pooling epilogue and the
        # prologue form of the pooled gradient, and no coordinate gradients (only the vote
        # aggregation, nsample = 16, asks for those).
        L_ = len(params) // 3
        compact = (_compact_enabled() and S >= 32 and S % 8 == 0 and L_ >= 2 and
                   params[3 * (L_ - 1)].shape[0] > 64 and _pool_in_epilogue() and
                   _pool_grad_in_prologue(S) and not ctx.needs_input_grad[0] and
                   not ctx.needs_input_grad[1] and
                   (not ctx.needs_input_grad[2] or N <= 8192) and
                   B * ((M + 63) // 64) <= 1024)   # the tile form of btr_sa_pool_bwd_coef
        _lib.btr_sac_bind(None)   # (a failed call must not leave this thread bound)
        cm = cplan = None
        with _on(xyz) as d:
            st = _stream(d)
            X0 = _f32((R, K0p), dev)
            if compact:
                groups = B * M
                i32 = dict(dtype=torch.int32, device=dev)
                cplan = dict(goff=torch.empty(groups + 1, **i32), dims=torch.empty(2, **i32),
                             cidx=torch.empty(R, **i32), bgrp=torch.empty(R // 8, **i32),
                             bw=_f32((R // 8,), dev))
                len_tmp = torch.empty(groups, **i32)
                _call(_lib.btr_sac_plan, groups, S, _p(idx), _p(len_tmp), _p(cplan["goff"]),
                      _p(cplan["dims"]), _p(cplan["cidx"]), _p(cplan["bgrp"]), _p(cplan["bw"]), st)
                cm = _make_cm(cplan, R)
                _call(_lib.btr_sac_gather, B, N, M, R, C, K0p, use_xyz, rdiv, _p(xyz),
                      _p(new_xyz), _p(feats_cl), _p(cplan["cidx"]), _p(cplan["bgrp"]),
                      _p(cplan["dims"]), _p(X0), st)
            else:
                _call(_lib.btr_sa_gather, B, N, M, S, C, K0p, use_xyz, rdiv, _p(xyz),
                      _p(new_xyz), _p(feats_cl), _p(idx), _p(X0), st)
            grid = _lib.btr_sa_gemm_grid(R)
            ext = None
            # shape key of the GEMM launches for the instrumented steps of bench.py: the rows
            # they really process (compact rows: the count is a device value, read back -- a
            # host sync -- only in that mode)
            Rk = int(cplan["dims"][0]) if compact and _ext.timing_detail() else R
            Ys, stats, Ws, counters = [], [], [], []
            A, lda, K = X0, K0p, K0p
            pa = pb = None
            bound = _ext.compact_bound(cm)
            bound.__enter__()
            # First-layer recompute (SA1: 4 input columns, nobody needs the input gradient):
            # the first pre-BN output is never stored; its consumers rebuild it from X0.
            rc = (K0p == 4 and L >= 3 and not any(ctx.needs_input_grad[0:3]) and
                  os.environ.get("BTR_SA_RECOMPUTE", "1") != "0")
            for l in range(L):
                W, gamma, beta = params[3 * l:3 * l + 3]
                Nl = W.shape[0]
                W2 = W.reshape(Nl, -1)
                if W2.shape[1] != K:  # layer 0 with a padded input width
                    Wp = torch.zeros((Nl, K), dtype=torch.float32, device=dev)
                    Wp[:, :W2.shape[1]] = W2
                    W2 = Wp
                W2 = W2.contiguous()
                part = _f32((grid, 2, Nl), dev)
                if rc and l == 0:    # statistics only
                    Y = _f32((0, Nl), dev)
                    _call(_lib.btr_sa_gemm_nt, R, Nl, K, _p(A), lda, _p(W2), K, None, Nl, None,
                          None, _p(part), st, key=(Rk, Nl, K, R))
                elif rc and l == 1:  # A = relu(bn(X0 . W0^T)) rebuilt while staging
                    Y = _f32((R, Nl), dev)
                    _call(_lib.btr_sa_gemm_nt_rc, R, Nl, K, _p(X0), _p(Ws[0]), _p(W2), K, _p(Y),
                          Nl, _p(pa), _p(pb), _p(part), st, key=(Rk, Nl, K, R))
                elif (l == L - 1 and pa is not None and _pool_in_epilogue() and
                      _lib.btr_sa_gemm_nt_poolfwd_supported(R, Nl, 8 if compact else S)):
                    # last layer: the GEMM epilogue also emits the per-group extrema (compact
                    # rows: per 8-row block), so the max-pool below does not read Y again
                    PSz = 8 if compact else S
                    Y = _f32((R, Nl), dev)
                    ext = (_f32((R // PSz, Nl), dev),
                           torch.empty((R // PSz, Nl), dtype=torch.uint8, device=dev))
                    _call(_lib.btr_sa_gemm_nt_poolfwd, R, Nl, K, _p(A), lda, _p(W2), K, _p(Y), Nl,
                          _p(pa), _p(pb), _p(part), PSz, _p(gamma), _p(ext[0]), _p(ext[1]), st,
                          key=(Rk, Nl, K, R))
                else:
                    Y = _f32((R, Nl), dev)
                    _call(_lib.btr_sa_gemm_nt, R, Nl, K, _p(A), lda, _p(W2), K, _p(Y), Nl,
                          _p(pa), _p(pb), _p(part), st, key=(Rk, Nl, K, R))
                scale, shift, mean, invstd = (_f32((Nl,), dev) for _ in range(4))
                bn = bns[l]
                if bn.momentum is None:
                    mom = 1.0 / float(bn.num_batches_tracked.item() + 1)
                else:
                    mom = float(bn.momentum)
                track = bn.track_running_stats and bn.running_mean is not None
                _call(_lib.btr_sa_bn_finalize, Nl, grid, float(R), float(bn.eps), mom, _p(part),
                      _p(gamma), _p(beta), _p(scale), _p(shift), _p(mean), _p(invstd),
                      _p(bn.running_mean if track else None),
                      _p(bn.running_var if track else None), st)
                if track:
                    counters.append(bn.num_batches_tracked)
                Ys.append(Y)
                Ws.append(W2)
                stats.append((scale, shift, mean, invstd))
                A, lda, K = Y, Nl, Nl
                pa, pb = scale, shift
            bound.__exit__()
            CL = Ys[-1].shape[1]
            out = _f32((B, CL, M), dev)
            out_cl = _f32((B, M, CL), dev)
            arg = torch.empty((B * M, CL), dtype=torch.uint8, device=dev)
            if compact:
                assert ext is not None
                _call(_lib.btr_sac_pool, B, M, CL, _p(ext[0]), _p(ext[1]), _p(cplan["goff"]),
                      _p(stats[-1][0]), _p(stats[-1][1]), _p(out), _p(out_cl), _p(arg), st)
            elif ext is not None:
                _call(_lib.btr_sa_pool_fin, B, M, CL, _p(ext[0]), _p(ext[1]), _p(stats[-1][0]),
                      _p(stats[-1][1]), _p(out), _p(out_cl), _p(arg), st)
            else:
                _call(_lib.btr_sa_pool, B, M, S, CL, CL, _p(Ys[-1]), _p(stats[-1][0]),
                      _p(stats[-1][1]), _p(out), _p(out_cl), _p(arg), st)
            if counters:  # one launch for the layer's num_batches_tracked += 1
                torch._foreach_add_(counters, 1)

        _ext.attach_twin(out, out_cl)  # lets the next fused layer skip a transpose
        ctx.dims = (B, N, M, S, C, use_xyz, rdiv, K0, K0p, L)
        ctx.rc = rc
        ctx.compact = compact
        ctx.pshapes = [p.shape for p in params]
        # save_for_backward (not attributes): saving the OUTPUT through an attribute would
        # create a ctx <-> out reference cycle that only the cyclic GC frees (GBs per step)
        flat_stats = [t for st4 in stats for t in st4]
        cp = [cplan[k] for k in ("goff", "dims", "cidx", "bgrp", "bw")] if compact else []
        ctx.save_for_backward(idx, X0, arg, out, *Ys, *Ws, *flat_stats, *cp)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, N, M, S, C, use_xyz, rdiv, K0, K0p, L = ctx.dims
        pshapes = ctx.pshapes
        saved = ctx.saved_tensors
        idx, X0, arg, out = saved[0:4]
        Ys = list(saved[4:4 + L])
        Ws = list(saved[4 + L:4 + 2 * L])
        flat = saved[4 + 2 * L:]
        stats = [tuple(flat[4 * l:4 * l + 4]) for l in range(L)]
        cm = cplan = None
        if ctx.compact:
            cplan = dict(zip(("goff", "dims", "cidx", "bgrp", "bw"), flat[4 * L:4 * L + 5]))
        dev = dout.device
        R = B * M * S
        dout = dout.contiguous()
        need_xyz, need_new, need_feat = ctx.needs_input_grad[0:3]
        grads = [None] * (3 * L)
        dxyz = dnew = dfeat = None

        _lib.btr_sac_bind(None)
        if cplan is not None:
            cm = _make_cm(cplan, R)
        with _on(dout) as d, _ext.compact_bound(cm):
            st = _stream(d)
            Rk = int(cplan["dims"][0]) if cplan is not None and _ext.timing_detail() else R
            # ---- last layer: max-pool + ReLU + BN backward.  Default: only the statistics
            # and the per-group coefficients are computed; the dense dY of the pooled layer
            # is formed inside the operand staging of its two GEMMs and never reaches HBM
            # (BTR_POOLGRAD=0: written in place over Y_last by btr_sa_pool_bwd).
            CL = Ys[-1].shape[1]
            scale, shift, mean, invstd = stats[-1]
            part = _f32((1024, 2, CL), dev)
            m1, m2, dg, db = (_f32((CL,), dev) for _ in range(4))
            pool = None
            if _pool_grad_in_prologue(S):
                dcl = _f32((B * M, CL), dev)
                alpha, beta = _f32((CL,), dev), _f32((CL,), dev)
                _call(_lib.btr_sa_pool_bwd_coef, B, M, S, CL, CL, _p(Ys[-1]), _p(dout), _p(out),
                      _p(arg), _p(mean), _p(invstd), _p(scale), _p(shift), _p(part), _p(m1),
                      _p(m2), _p(dg), _p(db), _p(dcl), _p(alpha), _p(beta), st)
                pool = (dcl, alpha, beta)
            else:
                _call(_lib.btr_sa_pool_bwd, B, M, S, CL, CL, _p(Ys[-1]), _p(dout), _p(out),
                      _p(arg), _p(mean), _p(invstd), _p(scale), _p(part), _p(m1), _p(m2),
                      _p(dg), _p(db), st)
            grads[3 * (L - 1) + 1], grads[3 * (L - 1) + 2] = dg, db
            dY = Ys[-1]
            # `lazy`: dY holds dZ_l (gradient w.r.t. layer l's activation) with BatchNorm_l's
            # backward sums in lazy = (m1, m2); btr_sa_bwd_fused applies them while it stages
            # its operand, any other consumer gets them applied in place first (the same
            # sequence as csrc/sa_layer.hip sa_layer_backward_add)
            lazy = None

            def fusable(j):
                # (the first-layer recompute rides in layer 1's fused call only behind <= 128
                # columns, as in csrc/sa_layer.hip)
                return j >= 1 and bool(_lib.btr_sa_bwd_fused_supported(
                    R, Ws[j].shape[0], Ws[j].shape[1])) and not (
                        ctx.rc and j == 1 and Ws[j].shape[0] > 128)
            for l in range(L - 1, -1, -1):
                Nl = dY.shape[1]
                W2 = Ws[l]
                K = W2.shape[1]
                if l == 0:
                    Xsrc, ldx, pa, pb = X0, K0p, None, None
                else:
                    Xsrc, ldx = Ys[l - 1], Ys[l - 1].shape[1]
                    pa, pb = stats[l - 1][0], stats[l - 1][1]
                pooled = pool is not None and l == L - 1  # dY is still Y_last + coefficients
                if ctx.rc and l == 0:
                    if lazy is not None:   # the recomputed first layer behind a fused call
                        sc, sh, mu, isd = stats[0]
                        C0 = Ws[0].shape[0]
                        nb = _lib.btr_sa_rc_wgrad_blocks(R, C0)
                        pw0 = _f32((nb, C0, 4), dev)
                        dW0 = _f32((C0, 4), dev)
                        _call(_lib.btr_sa_bn_relu_bwd_rc_apply, R, C0, C0, _p(dY), _p(X0),
                              _p(Ws[0]), _p(sc), _p(sh), _p(mu), _p(isd), _p(lazy[0]),
                              _p(lazy[1]), _p(pw0), _p(dW0), st)
                        grads[0] = dW0[:, :pshapes[0][1]].reshape(pshapes[0])
                    break  # (otherwise finished by btr_sa_bn_relu_bwd_rc below)
                if fusable(l) and (pooled or lazy is not None):
                    # the whole backward of layer l in one pass: dW_l, dZ_{l-1}, BatchNorm_{l-1}'s
                    # sums (csrc/sa_mlp.hip sa_bwd_fused_kernel)
                    chunks = _lib.btr_sa_bwd_fused_chunks(R, Nl, K)
                    pw = _f32((chunks, Nl, K), dev)
                    dW = _f32((Nl, K), dev)
                    Wt = W2.t().contiguous()  # (K, Nl)
                    G = _f32((R, K), dev)
                    part = _f32((chunks, 2, K), dev)
                    m1, m2, dg, db = (_f32((K,), dev) for _ in range(4))
                    scl, shl, mul, isl = stats[l]
                    scp, shp, mup, isp = stats[l - 1]
                    rc1 = ctx.rc and l == 1
                    _call(_lib.btr_sa_bwd_fused, R, Nl, K, _p(dY), Nl,
                          None if pooled else _p(Ys[l]), _p(scl), _p(shl), _p(mul), _p(isl),
                          None if pooled else _p(lazy[0]), None if pooled else _p(lazy[1]), S,
                          _p(arg) if pooled else None, _p(pool[0]) if pooled else None,
                          _p(pool[1]) if pooled else None, _p(pool[2]) if pooled else None,
                          _p(X0) if rc1 else _p(Xsrc), 4 if rc1 else ldx,
                          _p(Ws[0]) if rc1 else None, _p(pa), _p(pb), _p(mup), _p(isp), _p(Wt),
                          Nl, _p(G), K, _p(pw), _p(dW), _p(part), _p(m1), _p(m2), _p(dg), _p(db),
                          st, key=(Rk, Nl, K, R))
                    kin = pshapes[3 * l][1]
                    grads[3 * l] = dW[:, :kin].reshape(pshapes[3 * l])
                    grads[3 * (l - 1) + 1], grads[3 * (l - 1) + 2] = dg, db
                    dY = G
                    lazy = (m1, m2)
                    continue
                if lazy is not None:   # a consumer that wants dY_l itself
                    scl, shl, mul, isl = stats[l]
                    _call(_lib.btr_sa_bn_relu_bwd_apply, R, Nl, Nl, _p(dY), _p(Ys[l]), _p(scl),
                          _p(shl), _p(mul), _p(isl), _p(lazy[0]), _p(lazy[1]), st)
                    lazy = None
                # weight gradient: dW[n][k] = sum_r dY[r][n] * X_l[r][k]
                chunks = _lib.btr_sa_gemm_tn_chunks(R, Nl, K)
                pw = _f32((chunks, Nl, K), dev)
                dW = _f32((Nl, K), dev)
                if ctx.rc and l == 1:  # X = relu(bn(X0 . W0^T)) rebuilt while staging
                    _call(_lib.btr_sa_gemm_tn_rc, R, Nl, K, _p(dY), Nl, _p(X0), _p(Ws[0]),
                          _p(pa), _p(pb), _p(pw), _p(dW), st, key=(Rk, Nl, K, R))
                elif pooled:
                    _call(_lib.btr_sa_gemm_tn_pool, R, Nl, K, _p(dY), Nl, S, _p(arg),
                          _p(pool[0]), _p(pool[1]), _p(pool[2]), _p(Xsrc), ldx, _p(pa), _p(pb),
                          _p(pw), _p(dW), st, key=(Rk, Nl, K, R))
                else:
                    _call(_lib.btr_sa_gemm_tn, R, Nl, K, _p(dY), Nl, _p(Xsrc), ldx, _p(pa),
                          _p(pb), _p(pw), _p(dW), st, key=(Rk, Nl, K, R))
                kin = pshapes[3 * l][1]
                grads[3 * l] = dW[:, :kin].reshape(pshapes[3 * l])
                # input gradient: dX_l[r][k] = sum_n dY[r][n] * W[n][k]
                if l > 0 or need_xyz or need_new or need_feat:
                    Wt = W2.t().contiguous()  # (K, Nl)
                    # first layer, nobody asks for the coordinate part: only the feature columns
                    # of dX0, as a dense (R, C) tile (see csrc/sa_layer.hip feat_only)
                    feat_only = (l == 0 and not pooled and use_xyz and C > 0 and C % 4 == 0 and
                                 not (need_xyz and use_xyz) and not (need_new and use_xyz))
                    gld = C if feat_only else K
                    G = _f32((R, gld), dev)
                    if pooled:
                        _call(_lib.btr_sa_gemm_nt_pool, R, K, Nl, _p(dY), Nl, _p(Wt), Nl, _p(G),
                              K, S, _p(arg), _p(pool[0]), _p(pool[1]), _p(pool[2]), st,
                              key=(Rk, K, Nl, R))
                    elif feat_only:
                        _call(_lib.btr_sa_gemm_nt, R, C, Nl, _p(dY), Nl, Wt.data_ptr() + 12 * Nl,
                              Nl, _p(G), C, None, None, None, st, key=(Rk, C, Nl, R))
                    else:
                        _call(_lib.btr_sa_gemm_nt, R, K, Nl, _p(dY), Nl, _p(Wt), Nl, _p(G), K,
                              None, None, None, st, key=(Rk, K, Nl, R))
                    if l > 0:
                        sc, sh, mu, isd = stats[l - 1]
                        part = _f32((1024, 2, K), dev)
                        m1, m2, dg, db = (_f32((K,), dev) for _ in range(4))
                        if ctx.rc and l == 1:
                            # layer 0: BN+ReLU backward with y0 rebuilt from X0, fused with the
                            # layer's weight gradient -- dY0 is never written
                            nb = _lib.btr_sa_rc_wgrad_blocks(R, K)
                            pw0 = _f32((nb, K, 4), dev)
                            dW0 = _f32((K, 4), dev)
                            _call(_lib.btr_sa_bn_relu_bwd_rc, R, K, K, _p(G), _p(X0),
                                  _p(Ws[0]), _p(sc), _p(sh), _p(mu), _p(isd), _p(part), _p(m1),
                                  _p(m2), _p(dg), _p(db), _p(pw0), _p(dW0), st)
                            kin0 = pshapes[0][1]
                            grads[0] = dW0[:, :kin0].reshape(pshapes[0])
                        elif fusable(l - 1):   # the next layer applies the sums itself
                            _call(_lib.btr_sa_bn_relu_bwd_sums, R, K, K, _p(G), _p(Ys[l - 1]),
                                  _p(sc), _p(sh), _p(mu), _p(isd), _p(part), _p(m1), _p(m2),
                                  _p(dg), _p(db), st)
                            lazy = (m1, m2)
                        else:
                            _call(_lib.btr_sa_bn_relu_bwd, R, K, K, _p(G), _p(Ys[l - 1]),
                                  _p(sc), _p(sh), _p(mu), _p(isd), _p(part), _p(m1), _p(m2),
                                  _p(dg), _p(db), st)
                        grads[3 * (l - 1) + 1], grads[3 * (l - 1) + 2] = dg, db
                        dY = G
                    else:
                        dfeat_cl = None
                        if need_feat and C:
                            dfeat_cl = _f32((B, N, C), dev)
                        if need_xyz and use_xyz:
                            dxyz = _f32((B, N, 3), dev)
                        if need_new and use_xyz:
                            dnew = _f32((B, M, 3), dev)
                        if cplan is not None:   # compact rows: feature gradient only
                            wsb = _lib.btr_sac_scatter_workspace_bytes(B, N, R)
                            ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
                            if dfeat_cl is not None:
                                _call(_lib.btr_sac_scatter, B, N, M, C, gld,
                                      0 if feat_only else use_xyz, _p(G),
                                      _p(cplan["cidx"]), _p(cplan["goff"]), _p(dfeat_cl), _p(ws),
                                      wsb, R, st)
                        else:
                            wsb = _lib.btr_sa_scatter_workspace_bytes(B, N, M, S)
                            ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
                            _call(_lib.btr_sa_scatter, B, N, M, S, C, gld,
                                  0 if feat_only else use_xyz, rdiv, _p(G),
                                  _p(idx), _p(dfeat_cl), _p(dxyz), _p(dnew), _p(ws), wsb, st)
                        if dfeat_cl is not None:
                            dfeat = dfeat_cl.transpose(1, 2).contiguous()
        return (dxyz, dnew, dfeat, None, None) + tuple(grads)


# per module: {(shape, flags): (description, plan, gradient split sizes)}; weak, so that modules
# stay picklable / deep-copyable (ctypes structures with pointers are not)
_LAYER_CACHE = weakref.WeakKeyDictionary()


def native_enabled():
    """BTR_NATIVE_LAYERS=0: sequence the btr_sa_* launches from Python (FusedSAFunction, the
    readable statement of the sequence and the test oracle of csrc/sa_layer.hip) instead of
    one btr_sa_layer_forward / _backward call per layer.  The fully instrumented steps of
    bench.py (an event pair around every launch) also run the Python sequence."""
    return os.environ.get("BTR_NATIVE_LAYERS", "1") != "0" and not _ext.timing_detail()


def _sa_options():
    o = 0
    if _compact_enabled():
        o |= _ext.SA_OPT_COMPACT
    if os.environ.get("BTR_SA_RECOMPUTE", "1") != "0":
        o |= _ext.SA_OPT_RECOMPUTE
    if _pool_in_epilogue():
        o |= _ext.SA_OPT_POOL_EPILOGUE
    if os.environ.get("BTR_POOLGRAD", "1") != "0":
        o |= _ext.SA_OPT_POOL_GRAD
    # the pooled layer without its stored pre-BN output (btr_sa_bwd_gram; whole-layer / whole-
    # backbone calls only: the Python-sequenced FusedSAFunction keeps the Y_l-reading form).
    # The environment switches of the kernels it rests on are part of the option word, so that a
    # plan cached under other settings is not reused
    if (os.environ.get("BTR_POOL_GRAM", "1") != "0" and
            os.environ.get("BTR_FWD_STREAM", "1") != "0" and
            os.environ.get("BTR_BWD_FUSED", "1") != "0" and
            os.environ.get("BTR_GEMM", "") != "f32"):
        o |= _ext.SA_OPT_POOL_GRAM
    # per-point first layer (whole-layer / whole-backbone calls; BTR_SA_PPFL=0: row-wise)
    if os.environ.get("BTR_SA_PPFL", "1") != "0" and os.environ.get("BTR_GEMM", "") != "f32":
        o |= _ext.SA_OPT_PPFL
        if os.environ.get("BTR_SA_PPFL_XYZ") == "1":   # (vote aggregation too: measured neutral)
            o |= _ext.SA_OPT_PPFL_XYZ
    return o


def _u8(nbytes, dev):
    return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=dev)


class FusedSALayer(Function):
    """FusedSAFunction with the launch sequence in C++ (btr_sa_layer_forward / _backward,
    csrc/sa_layer.hip): per layer one description, one plan (cached per shape), three buffers
    (saved / scratch / flat gradients) and one call each way.  Same kernels, same results."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx, meta, sink, *params):
        _ext.RUNNING_STATS_EPOCH[0] += 1   # running statistics move through raw pointers
        dev = xyz.device
        B, N, _ = xyz.shape
        M, S = idx.shape[1], idx.shape[2]
        C = features.shape[1] if features is not None else 0
        bns = meta["bns"]
        L = len(params) // 3
        ctx.to_sink = sink is not None
        xyz = xyz.contiguous()
        new_xyz = new_xyz.contiguous()
        feats_cl = None
        if C:
            feats_cl = _ext.twin_of(features)
            if os.environ.get("BTR_SA_CL_SHORTCUT", "1") == "0":
                feats_cl = None
            if feats_cl is None or feats_cl.shape != (B, N, C):
                feats_cl = features.transpose(1, 2).contiguous()
        ent = _sa_entry(meta, B, N, M, S, C, ctx.needs_input_grad, params)
        d, plan, sizes = ent[:3]
        for l in range(L):
            W, gamma, beta = params[3 * l:3 * l + 3]
            bn = bns[l]
            assert W.is_contiguous()
            d.w[l], d.gamma[l], d.beta[l] = W.data_ptr(), gamma.data_ptr(), beta.data_ptr()
            track = bn.track_running_stats and bn.running_mean is not None
            d.running_mean[l] = bn.running_mean.data_ptr() if track else None
            d.running_var[l] = bn.running_var.data_ptr() if track else None
            d.num_batches_tracked[l] = bn.num_batches_tracked.data_ptr() if track else None
            d.momentum[l] = float(bn.momentum) if bn.momentum is not None else \
                1.0 / float(bn.num_batches_tracked.item() + 1)
        CL = d.width[L - 1]
        out = _f32((B, CL, M), dev)
        out_cl = _f32((B, M, CL), dev)
        saved = _u8(plan.saved_bytes, dev)
        scratch = _u8(plan.fwd_scratch_bytes, dev)
        with _on(xyz) as dv:
            _call(_lib.btr_sa_layer_forward, ctypes.addressof(d), ctypes.addressof(plan),
                  _p(xyz), _p(new_xyz), _p(feats_cl), _p(idx), _p(out), _p(out_cl), _p(saved),
                  _p(scratch), _stream(dv))
        _ext.attach_twin(out, out_cl)
        ctx.plan = tuple(ent[:3])
        ctx.dims = (B, N, M, C)
        ctx.pshapes = [p.shape for p in params]
        ctx.save_for_backward(idx, saved, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        d, plan, sizes = ctx.plan
        B, N, M, C = ctx.dims
        idx, saved, out = ctx.saved_tensors
        dev = dout.device
        dout = dout.contiguous()
        grads = _f32((plan.grads_floats,), dev)
        scratch = _u8(plan.bwd_scratch_bytes, dev)
        dfeat = _f32((B, C, N), dev) if d.need_dfeat and C else None
        dxyz = _f32((B, N, 3), dev) if d.need_dxyz and d.use_xyz else None
        dnew = _f32((B, M, 3), dev) if d.need_dnew_xyz and d.use_xyz else None
        with _on(dout) as dv:
            _call(_lib.btr_sa_layer_backward, ctypes.addressof(d), ctypes.addressof(plan),
                  _p(idx), _p(out), _p(dout), _p(saved), _p(grads), _p(dfeat), _p(dxyz),
                  _p(dnew), _p(scratch), _stream(dv))
        if ctx.to_sink:   # (one flat gradient for the sink, grad_sink.py)
            return (dxyz, dnew, dfeat, None, None, grads) + (None,) * len(ctx.pshapes)
        return (dxyz, dnew, dfeat, None, None, None) + tuple(
            sa_grad_views(d, plan, sizes, ctx.pshapes, grads))


def _sa_entry(meta, B, N, M, S, C, need, params):
    """[description, plan, gradient block sizes, sink | None] of a fused SA layer at this shape
    (`need`: which of xyz / new_xyz / features take a gradient)."""
    key = (B, N, M, S, C, bool(need[0]), bool(need[1]), bool(need[2]), _sa_options())
    cache = meta["cache"]
    ent = cache.get(key)
    if ent is None:
        bns = meta["bns"]
        L = len(params) // 3
        d = _ext.SaLayer()
        d.b, d.n, d.m, d.s, d.c = B, N, M, S, C
        d.use_xyz = 1 if meta["use_xyz"] else 0
        d.radius_div = float(meta["radius_div"])
        d.layers = L
        for l in range(L):
            d.width[l] = params[3 * l].shape[0]
            d.eps[l] = float(bns[l].eps)
        d.need_dxyz, d.need_dnew_xyz, d.need_dfeat = int(need[0]), int(need[1]), int(need[2])
        d.options = key[-1]
        plan = _ext.SaPlan()
        _call(_lib.btr_sa_layer_plan, ctypes.addressof(d), ctypes.addressof(plan))
        sizes = []
        for l in range(L):
            sizes += [d.width[l] * plan.kin[l], d.width[l], d.width[l]]
        ent = cache[key] = [d, plan, sizes, None]
    return ent


def _sa_sink(meta, xyz, new_xyz, features, idx, params):
    """The flat gradient sink of this layer call (grad_sink.py), or None outside its scope."""
    if not grad_sink.active() or not grad_sink.all_leaves(params):
        return None
    B, N, _ = xyz.shape
    need = (xyz.requires_grad, new_xyz.requires_grad,
            features is not None and features.requires_grad)
    ent = _sa_entry(meta, B, N, idx.shape[1], idx.shape[2],
                    features.shape[1] if features is not None else 0, need, params)
    if ent[3] is None or ent[3].tensor.device != xyz.device:
        d, plan, sizes = ent[:3]
        pshapes = [p.shape for p in params]
        ent[3] = grad_sink.Sink(plan.grads_floats, xyz.device,
                                lambda g: sa_grad_views(d, plan, sizes, pshapes, g))
    return ent[3].bind(params)


def sa_grad_views(d, plan, sizes, pshapes, grads):
    """[dW, dgamma, dbeta] per layer as views of a fused SA layer's flat gradient buffer."""
    parts = grads.split(sizes)
    res = []
    for l in range(d.layers):
        shape = pshapes[3 * l]
        if l == 0 and not plan.recompute:
            # written without the padding columns (csrc/sa_layer.hip reduce_unpad_next)
            dW = parts[0][:d.width[0] * shape[1]].view(d.width[0], shape[1])
        else:
            dW = parts[3 * l].view(d.width[l], plan.kin[l])
            if plan.kin[l] != shape[1]:
                dW = dW[:, :shape[1]]
        res += [dW.reshape(shape), parts[3 * l + 1], parts[3 * l + 2]]
    return res


def _eval_constants(module, k0):
    """[(W (width, k_in) with layer 0 padded to k0 columns (k0 None: as it is), a, b)] per MLP layer of a
    set-abstraction module in inference mode, where BatchNorm with running statistics is the
    affine map  y_bn = (y - running_mean) / sqrt(running_var + eps) * gamma + beta = a*y + b.
    Derived once per evaluation pass instead of with ~7 small torch launches per layer and call
    (15 layers x every batch: the inference forward is launch-paced).  The
    cache lives on the module, keyed by the tensors' version counters (in-place loads such as
    load_state_dict) and by _ext.RUNNING_STATS_EPOCH (the library's own writes through raw
    pointers, which version counters do not see); every train()/eval() switch drops it
    (_SingleScaleSA.train)."""
    layers = list(module.mlp_module)
    tensors = []
    for l in layers:
        bn = l.bn.bn
        tensors += [l.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    key = (k0, _ext.RUNNING_STATS_EPOCH[0]) + tuple((t.data_ptr(), t._version) for t in tensors)
    hit = getattr(module, '_btr_eval_consts', None)
    if hit is not None and hit[0] == key:
        return hit[1]
    res, K = [], k0
    with torch.no_grad():
        for l in layers:
            W, bn = l.conv.weight, l.bn.bn
            W2 = W.reshape(W.shape[0], -1)
            if K is not None and W2.shape[1] != K:  # layer 0 with a padded input width
                Wp = torch.zeros((W.shape[0], K), dtype=torch.float32, device=W.device)
                Wp[:, :W2.shape[1]] = W2
                W2 = Wp
            a = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
            res.append((W2.contiguous(), a, (bn.bias - bn.running_mean * a).contiguous()))
            K = W.shape[0]
    module._btr_eval_consts = (key, res)
    return res


def fused_eval_forward(module, xyz, new_xyz, features, idx):
    """Inference-mode forward of a set-abstraction layer (module.eval(), under no_grad -- the
    evaluation pass of the reference, train_Votenet_FSB.py:246-293): BatchNorm uses its RUNNING
    statistics, so no batch reduction separates the layers and nothing is saved for a backward:
    gather -> one GEMM per layer with the previous layer's BN + ReLU applied while its operand
    is staged -> BN + ReLU + max-pool.  Same kernels as the training forward, minus the
    statistics epilogues / finalize launches; the grouped tensor never exists."""
    g = module.grouper
    dev = xyz.device
    B, N, _ = xyz.shape
    M, S = idx.shape[1], idx.shape[2]
    C = features.shape[1] if features is not None else 0
    use_xyz = 1 if g.use_xyz else 0
    rdiv = float(g.radius if g.normalize_xyz else 1.0)
    R = B * M * S
    K0p = _ceil4(3 * use_xyz + C)
    xyz = xyz.contiguous()
    new_xyz = new_xyz.contiguous()
    feats_cl = None
    if C:
        feats_cl = _ext.twin_of(features)
        if feats_cl is None or feats_cl.shape != (B, N, C):
            feats_cl = features.transpose(1, 2).contiguous()
    layers = list(module.mlp_module)
    widths = [l.conv.weight.shape[0] for l in layers]
    if len(layers) == 3 and _lib.btr_sa_eval_fused_supported(S, C, use_xyz, *widths):
        # the whole layer as ONE launch (csrc/sa_mlp.hip sa_eval_fused_kernel): no rows x channels
        # tensor reaches HBM
        consts = _eval_constants(module, 4)
        ws = [c[0] for c in consts]
        ab = [t for c in consts for t in c[1:]]
        out = _f32((B, widths[2], M), dev)
        out_cl = _f32((B, M, widths[2]), dev)
        with _on(xyz) as d:
            _call(_lib.btr_sa_eval_fused, B, N, M, S, C, use_xyz, rdiv, _p(xyz), _p(new_xyz),
                  _p(feats_cl), _p(idx.contiguous()), widths[0], widths[1], widths[2], _p(ws[0]),
                  _p(ws[1]), widths[0], _p(ws[2]), widths[1], *[_p(t) for t in ab], _p(out),
                  _p(out_cl), _stream(d))
        _ext.attach_twin(out, out_cl)
        return out
    if native_enabled() and len(layers) <= _ext.MAX_LAYERS and \
            all(w % 4 == 0 for w in widths) and idx.dtype == torch.int32:
        # the training forward's kernels (compact rows, per-point first layer, streaming GEMMs,
        # pool in the last GEMM's epilogue) with the running-statistics affine map handed in and
        # no finaliser: ONE library call (csrc/sa_layer.hip, BTR_SA_OPT_EVAL)
        return _native_eval_forward(module, xyz, new_xyz, feats_cl, idx, B, N, M, S, C, widths)
    with _on(xyz) as d:
        st = _stream(d)
        A = _f32((R, K0p), dev)
        _call(_lib.btr_sa_gather, B, N, M, S, C, K0p, use_xyz, rdiv, _p(xyz), _p(new_xyz),
              _p(feats_cl), _p(idx), _p(A), st)
        lda, K, pa, pb = K0p, K0p, None, None
        for W2, na, nb in _eval_constants(module, K0p):
            Nl = W2.shape[0]
            Y = _f32((R, Nl), dev)
            _call(_lib.btr_sa_gemm_nt, R, Nl, K, _p(A), lda, _p(W2), K, _p(Y), Nl, _p(pa), _p(pb),
                  None, st, key=(R, Nl, K))
            pa, pb = na, nb   # applied while the next launch stages this layer's output
            A, lda, K = Y, Nl, Nl
        CL = K
        out = _f32((B, CL, M), dev)
        out_cl = _f32((B, M, CL), dev)
        arg = torch.empty((B * M, CL), dtype=torch.uint8, device=dev)
        _call(_lib.btr_sa_pool, B, M, S, CL, CL, _p(A), _p(pa), _p(pb),
              _p(out), _p(out_cl), _p(arg), st)
    _ext.attach_twin(out, out_cl)
    return out


def _native_eval_forward(module, xyz, new_xyz, feats_cl, idx, B, N, M, S, C, widths):
    g = module.grouper
    dev = xyz.device
    consts = _eval_constants(module, None)
    opts = _sa_options() | _ext.SA_OPT_EVAL
    cache = module.__dict__.setdefault('_btr_eval_plans', {})
    key = (B, N, M, S, C, opts)
    ent = cache.get(key)
    L = len(widths)
    if ent is None:
        d = _ext.SaLayer()
        d.b, d.n, d.m, d.s, d.c = B, N, M, S, C
        d.use_xyz = 1 if g.use_xyz else 0
        d.radius_div = float(g.radius if g.normalize_xyz else 1.0)
        d.layers = L
        for l in range(L):
            d.width[l] = widths[l]
            d.eps[l] = float(module.mlp_module[l].bn.bn.eps)
        d.options = opts
        plan = _ext.SaPlan()
        _call(_lib.btr_sa_layer_plan, ctypes.addressof(d), ctypes.addressof(plan))
        # (saved / scratch live with the module: an evaluation pass calls every layer once per
        # batch, nothing is kept for a backward)
        ent = cache[key] = [d, plan, _u8(plan.saved_bytes, dev), _u8(plan.fwd_scratch_bytes, dev)]
    d, plan, saved, scratch = ent
    if saved.device != dev:
        saved, scratch = ent[2], ent[3] = _u8(plan.saved_bytes, dev), _u8(plan.fwd_scratch_bytes, dev)
    for l in range(L):
        W, a, b = consts[l]
        d.w[l], d.gamma[l], d.beta[l] = W.data_ptr(), a.data_ptr(), b.data_ptr()
        d.running_mean[l] = d.running_var[l] = d.num_batches_tracked[l] = None
    CL = widths[-1]
    out = _f32((B, CL, M), dev)
    out_cl = _f32((B, M, CL), dev)
    with _on(xyz) as dv:
        _call(_lib.btr_sa_layer_forward, ctypes.addressof(d), ctypes.addressof(plan), _p(xyz),
              _p(new_xyz), _p(feats_cl), _p(idx.contiguous()), _p(out), _p(out_cl), _p(saved),
              _p(scratch), _stream(dv))
    _ext.attach_twin(out, out_cl)
    return out


def _compact_enabled():
    """BTR_SA_COMPACT=0: evaluate every layer on all nsample rows per group (dense rows)."""
    return os.environ.get("BTR_SA_COMPACT", "1") != "0"


def _make_cm(cplan, dense_rows):
    cm = _ext.CompactRows(_p(cplan["dims"]), _p(cplan["bw"]), _p(cplan["bgrp"]), _p(cplan["goff"]),
                          float(dense_rows))
    cm._keep = cplan   # the tensors stay alive as long as the description does
    return cm


def _pool_in_epilogue():
    """BTR_POOL_EPILOGUE=0: the max-pool reads the last layer's output again (btr_sa_pool)
    instead of using the extrema the GEMM epilogue emits."""
    return os.environ.get("BTR_POOL_EPILOGUE", "1") != "0"


def _pool_grad_in_prologue(nsample):
    """The GEMM-prologue form of the pooled layer's gradient needs whole groups per staged
    tile (128 rows in the NT kernel, 32 in the TN kernel)."""
    return nsample in (16, 32, 64, 128) and os.environ.get("BTR_POOLGRAD", "1") != "0"


def can_fuse(module, xyz, features):
    """True when `module` (a _SingleScaleSA) is in the configuration the fused kernels cover:
    ball-query grouping, max-pool, every MLP layer = bias-free 1x1 conv + BatchNorm2d + ReLU,
    CUDA tensors; training mode (batch statistics, forward + backward) or inference mode under
    no_grad (running statistics, forward only: fused_eval_forward)."""
    import torch.nn as nn
    if not (enabled() and xyz.is_cuda and module.pooling == 'max'):
        return False
    if not module.training and torch.is_grad_enabled():
        return False   # eval-mode layers that still need a backward: op-by-op path
    g = module.grouper
    if not isinstance(g, pointnet2_utils.QueryAndGroup) or g.sample_uniformly:
        return False
    if g.nsample > 255 or len(module.mlp_module) == 0:
        return False
    for layer in module.mlp_module:
        names = [n for n, _ in layer.named_children()]
        if names != ["conv", "bn", "activation"]:
            return False
        if layer.conv.bias is not None or layer.conv.kernel_size != (1, 1):
            return False
        if not isinstance(layer.activation, nn.ReLU) or not isinstance(layer.bn.bn, nn.BatchNorm2d):
            return False
        if layer.bn.bn.weight is None:
            return False
        if not module.training and (layer.bn.bn.training or layer.bn.bn.running_mean is None):
            return False
    return features is None or features.dtype == torch.float32


def fused_group_mlp_max(module, xyz, new_xyz, features):
    """Drop-in for grouper + mlp_module + max-pool of a _SingleScaleSA module."""
    g = module.grouper
    # (computed with the prefetched pyramid; void once xyz / new_xyz were written to in place)
    pre = _ext.derived(new_xyz, "_btr_ball_query", xyz)
    if pre is not None and pre[1] is xyz and pre[2] == g.radius and pre[3] == g.nsample:
        idx = pre[0]
    else:
        idx = pointnet2_utils.ball_query(g.radius, g.nsample, xyz, new_xyz)
    if features is None and not g.use_xyz:
        raise AssertionError("Cannot have not features and not use xyz as a feature!")
    if not module.training:
        return fused_eval_forward(module, xyz, new_xyz, features, idx)
    params = []
    bns = []
    for layer in module.mlp_module:
        params += [layer.conv.weight, layer.bn.bn.weight, layer.bn.bn.bias]
        bns.append(layer.bn.bn)
    cache = _LAYER_CACHE.get(module)
    if cache is None:
        cache = _LAYER_CACHE[module] = {}
    meta = {"radius_div": g.radius if g.normalize_xyz else 1.0, "use_xyz": g.use_xyz, "bns": bns,
            "cache": cache}
    if native_enabled() and len(bns) <= _ext.MAX_LAYERS:
        return FusedSALayer.apply(xyz, new_xyz, features, idx, meta,
                                  _sa_sink(meta, xyz, new_xyz, features, idx, params), *params)
    return FusedSAFunction.apply(xyz, new_xyz, features, idx, meta, *params)
