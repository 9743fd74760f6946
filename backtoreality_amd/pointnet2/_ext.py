"""`pointnet2._ext` for MI355X: the nine callables of the reference's pybind module
(/root/reference/detection/Votenet/pointnet2/_ext_src/src/bindings.cpp:11-24), implemented as a
thin ctypes shim over the C ABI of ``libbtr_pointnet2.so`` (include/btr_pointnet2.h).

The shim does what the reference's C++ wrappers do (src/{sampling,ball_query,group_points,
interpolate}.cpp): check contiguity / dtype / device (utils.h:10-30 -> RuntimeError), allocate
the result on the input's device, and enqueue the kernel on the CURRENT stream of that device
without synchronising.  CPU tensors raise ``RuntimeError("CPU not supported")`` exactly like
the reference (e.g. ball_query.cpp:33): there is no CPU fallback in the product path, and a
missing HIP library is an ImportError at import time, never a silent downgrade.
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# one library per rounding mode of the squared distance (include/btr_pointnet2.h,
# btr_distance_mode): 1 = nvcc-style contraction (default), 0 = as written, 2 = left-to-right
_LIB_NAMES = {1: "libbtr_pointnet2.so", 0: "libbtr_pointnet2_fmad0.so",
              2: "libbtr_pointnet2_fmad2.so"}
LIB_PATH = os.path.join(_PKG, "lib", _LIB_NAMES[1])

_vp = ctypes.c_void_p
_ci = ctypes.c_int
_cf = ctypes.c_float
_cd = ctypes.c_double
_ll = ctypes.c_longlong
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/btr_pointnet2.h one to one.
_SIGNATURES = {
    "btr_abi_version": (_ci, []),
    "btr_build_id": (ctypes.c_char_p, []),
    "btr_distance_mode": (_ci, []),
    "btr_last_error": (ctypes.c_char_p, []),
    "btr_opt_n_threads": (_ci, [_ci]),
    "btr_furthest_point_sampling": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp]),
    "btr_furthest_point_sampling_bs": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp]),
    "btr_furthest_point_sampling_workspace_bytes": (_sz, [_ci, _ci, _ci]),
    "btr_furthest_point_sampling_ws": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _sz, _vp]),
    "btr_backbone_fork_event": (None, [_vp, _ci]),
    "btr_grid_cus": (_ci, []),
    "btr_fps_lds_reserve_kb": (_ci, []),
    "btr_fps_lds_kb": (_ci, [_ci]),
    "btr_fps_set_lds_kb": (None, [_ci]),
    "btr_fps_ordered_scratch_bytes": (_sz, [_ci, _ci, _ci]),
    "btr_furthest_point_sampling_ordered": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _sz,
                                                  _vp]),
    "btr_gather_points": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]),
    "btr_gather_points_grad": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]),
    "btr_ball_query": (_ci, [_ci, _ci, _ci, _cf, _ci, _vp, _vp, _vp, _vp]),
    "btr_ball_query_workspace_bytes": (_sz, [_ci, _ci, _ci, _ci]),
    "btr_ball_query_ws": (_ci, [_ci, _ci, _ci, _cf, _ci, _vp, _vp, _vp, _vp, _sz, _vp]),
    "btr_ball_query_buckets_workspace_bytes": (_sz, [_ci, _ci, _ci, _ci]),
    "btr_ball_query_buckets": (_ci, [_ci, _ci, _ci, _cf, _ci, _vp, _vp, _vp, _vp, _sz, _vp]),
    "btr_group_points": (_ci, [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]),
    "btr_group_points_grad": (_ci, [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]),
    "btr_three_nn": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]),
    "btr_three_nn_weights": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    "btr_three_interpolate": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]),
    "btr_three_interpolate_grad": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]),
    # fused set-abstraction MLP (used by fused_sa.py)
    "btr_sa_gather": (_ci, [_ci, _ci, _ci, _ci, _ci, _ci, _ci, _cf, _vp, _vp, _vp, _vp, _vp, _vp]),
    "btr_sa_gemm_grid": (_ci, [_ci]),
    "btr_sa_gemm_nt": (_ci, [_ci, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp]),
    "btr_sa_bn_finalize": (_ci, [_ci, _ci, _cd, _cf, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp]),
    "btr_sa_pool": (_ci, [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "btr_sa_pool_bwd": (_ci, [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp, _vp, _vp]),
    "btr_sa_bn_relu_bwd": (_ci, [_ll, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp]),
    "btr_sa_gemm_tn_chunks": (_ci, [_ci, _ci, _ci]),
    "btr_sa_gemm_tn": (_ci, [_ci, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp]),
    "btr_sa_scatter_workspace_bytes": (_sz, [_ci, _ci, _ci, _ci]),
    "btr_sa_scatter": (_ci, [_ci, _ci, _ci, _ci, _ci, _ci, _ci, _cf, _vp, _vp, _vp, _vp, _vp,
                             _vp, _sz, _vp]),
    "btr_sa_gemm_nt_rc": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp]),
    "btr_sa_gemm_tn_rc": (_ci, [_ci, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "btr_sa_rc_wgrad_blocks": (_ci, [_ll, _ci]),
    "btr_sa_bn_relu_bwd_rc": (_ci, [_ll, _ci, _ci] + [_vp] * 15),
    "btr_sa_pool_bwd_coef": (_ci, [_ci, _ci, _ci, _ci, _ci] + [_vp] * 17),
    "btr_sa_eval_fused_supported": (_ci, [_ci] * 6),
    "btr_sa_eval_fused": (_ci, [_ci] * 6 + [_cf] + [_vp] * 4 + [_ci] * 3 + [_vp, _vp, _ci, _vp, _ci] +
                          [_vp] * 9),
    "btr_sa_bwd_fused_supported": (_ci, [_ci, _ci, _ci]),
    "btr_sa_bwd_fused_chunks": (_ci, [_ci, _ci, _ci]),
    "btr_sa_bwd_fused": (_ci, [_ci, _ci, _ci, _vp, _ci] + [_vp] * 7 + [_ci] + [_vp] * 5 +
                         [_ci] + [_vp] * 6 + [_ci, _vp, _ci] + [_vp] * 8),
    "btr_sa_bwd_gram_supported": (_ci, [_ci, _ci, _ci]),
    "btr_sa_bwd_gram_chunks": (_ci, [_ci, _ci, _ci]),
    "btr_sa_bwd_gram_scratch_floats": (_sz, [_ci, _ci, _ci]),
    "btr_sa_bwd_gram": (_ci, [_ci, _ci, _ci, _vp, _ci] + [_vp] * 6 + [_ci, _ci] + [_vp] * 5 +
                        [_ci] + [_vp] * 9),
    "btr_sa_bn_relu_bwd_rc_apply": (_ci, [_ll, _ci, _ci] + [_vp] * 12),
    "btr_sa_bn_relu_bwd_sums": (_ci, [_ll, _ci, _ci] + [_vp] * 12),
    "btr_sa_bn_relu_bwd_apply": (_ci, [_ll, _ci, _ci] + [_vp] * 9),
    "btr_sa_gemm_nt_pool": (_ci, [_ci, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _ci, _vp, _vp,
                                  _vp, _vp, _vp]),
    "btr_sa_gemm_tn_pool": (_ci, [_ci, _ci, _ci, _vp, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _ci,
                                  _vp, _vp, _vp, _vp, _vp]),
    # compact rows of the fused set-abstraction path (used by fused_sa.py)
    "btr_sac_bind": (None, [_vp]),
    "btr_sac_plan": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "btr_sac_gather": (_ci, [_ci, _ci, _ci, _ci, _ci, _ci, _ci, _cf] + [_vp] * 8),
    "btr_sac_pool": (_ci, [_ci, _ci, _ci] + [_vp] * 9),
    "btr_sac_pool_y": (_ci, [_ci, _ci, _ci] + [_vp] * 10),
    "btr_sa_pool_fin_y": (_ci, [_ci, _ci, _ci] + [_vp] * 9),
    "btr_sa_gemm_nt_poolfwd_nostore_supported": (_ci, [_ci, _ci, _ci, _ci]),
    "btr_sac_scatter_workspace_bytes": (_sz, [_ci, _ci, _ci]),
    "btr_sac_scatter": (_ci, [_ci, _ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _sz, _ci,
                              _vp]),
    # point-wise MLP chains (used by fused_mlp.py)
    "btr_pm_gemm_grid": (_ci, [_ci]),
    "btr_pm_weight_planes_bytes": (_sz, [_ci, _ci]),
    "btr_pm_weight_planes": (_ci, [_ci, _ci, _vp, _ci, _vp, _vp]),
    "btr_pm_gemm_nt_sm_supported": (_ci, [_ci, _ci, _ci]),
    "btr_pm_gemm_nt_sm": (_ci, [_ci, _ci, _ci, _vp, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _vp]),
    "btr_pm_gemm_nt": (_ci, [_ci, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp]),
    "btr_pm_out": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _vp, _vp]),
    "btr_pm_rows": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp]),
    "btr_fps_time_next_kernel": (None, [_vp, _vp]),
    "btr_ball_query_time_next": (None, [_vp, _vp]),
    "btr_vote_assemble": (_ci, [_ci, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "btr_vote_assemble_bwd": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    # fused attention core (groupfree/fused_attention.py)
    "btr_attention_supported": (_ci, [_ci]),
    "btr_attention_fwd": (_ci, [_ci] * 5 + [_vp, _ll, _ll, _vp, _vp, _ll, _ll, _vp, _vp, _cf, _cf,
                                            ctypes.c_ulonglong, _vp, _vp]),
    "btr_attention_bwd": (_ci, [_ci] * 5 + [_vp, _ll, _ll, _vp, _vp, _ll, _ll, _vp, _vp, _vp, _vp,
                                            _vp, _ll, _ll, _vp, _vp, _ll, _ll, _cf, _cf,
                                            ctypes.c_ulonglong, _vp, _vp]),
    # whole-layer entry points (csrc/sa_layer.hip): description + plan by address
    "btr_sa_layer_plan": (_ci, [_vp, _vp]),
    "btr_sa_layer_ppfl": (_ci, [_vp, _vp]),
    "btr_sa_layer_forward": (_ci, [_vp] * 11),
    "btr_sa_layer_backward": (_ci, [_vp] * 12),
    "btr_pm_chain_plan": (_ci, [_vp, _vp]),
    "btr_pm_chain_forward": (_ci, [_vp] * 9),
    "btr_pm_chain_backward": (_ci, [_vp] * 9),
    # whole-backbone entry points (csrc/backbone.hip): description + plan by address
    "btr_backbone_plan": (_ci, [_vp, _vp]),
    "btr_backbone_sampling": (_ci, [_vp] * 6),
    "btr_backbone_forward": (_ci, [_vp] * 7 + [_ci, _vp]),
    "btr_backbone_backward": (_ci, [_vp] * 10),
    # whole decoder layer (csrc/decoder.hip, used by groupfree/fused_decoder.py)
    "btr_decoder_layer_plan": (_ci, [_vp, _vp]),
    "btr_decoder_layer_forward": (_ci, [_vp] * 11),
    "btr_decoder_layer_backward": (_ci, [_vp] * 14),
    # GroupFree3D per-head loss (csrc/gf_loss.hip, used by groupfree/fused_loss.py)
    "btr_gf_loss_part_floats": (_ci, [_ci, _ci, _ci]),
    "btr_gf_loss_fwd": (_ci, [_vp] * 21),
    "btr_gf_loss_weak_fwd": (_ci, [_vp] * 14),
    "btr_focal_sum": (_ci, [_ci, _ci, _ci, _vp, _vp] + [ctypes.c_float] * 4 + [_vp, _vp, _vp]),
    "btr_gf_head_decode": (_ci, [_ci] * 4 + [_vp] + [ctypes.c_longlong] * 3 + [_vp] * 9),
    "btr_gemm_trace_begin": (None, []),
    "btr_gemm_trace_end": (_ci, [_vp, _vp]),
    "btr_gemm_trace_work": (_ci, [_vp, _vp, _vp]),
    "btr_gf_stack_sizeof": (ctypes.c_longlong, [_ci]),
    "btr_graph_stats": (None, [_vp] * 3),
    "btr_graph_clear": (None, []),
    "btr_gf_stack_plan": (_ci, [_vp, _vp]),
    "btr_gf_stack_forward": (_ci, [_vp] * 21),
    "btr_gf_stack_backward": (_ci, [_vp] * 12),
    # multi-tensor Adam / AdamW (csrc/optimizer.hip, used by votenet/train.py)
    "btr_adam_chunk": (_ci, []),
    "btr_adam_multi": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double,
                             ctypes.c_double, _ci, _vp, _vp]),
    "btr_grad_sumsq_multi": (_ci, [_ci, _ci, _vp, _vp, _vp, _vp, _vp]),
    "btr_grad_norm_final": (_ci, [_ci, _vp, ctypes.c_float, _vp, _vp]),
    # fused VoteNet loss (used by votenet/fused_loss.py)
    "btr_domain_loss": (_ci, [_ci, _ci, _cf, _cf] + [_vp] * 9),
    "btr_votenet_loss_fwd": (_ci, [_ci] * 9 + [_vp] * 24 + [_vp, _ci, _vp, _vp]),
    "btr_votenet_loss_bwd": (_ci, [_ci] * 9 + [_vp] * 26 + [_vp, _ci, _vp, _vp]),
    "btr_gather_rows": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]),
    "btr_gather_rows_grad": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]),
    "btr_sa_gemm_nt_poolfwd_supported": (_ci, [_ci, _ci, _ci]),
    "btr_sa_gemm_nt_poolfwd": (_ci, [_ci, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp,
                                     _ci, _vp, _vp, _vp, _vp]),
    "btr_sa_pool_fin": (_ci, [_ci, _ci, _ci] + [_vp] * 8),
    # evaluation-side box arithmetic (used by votenet/ap_helper.py)
    "btr_nms_boxes": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp, ctypes.c_double, _ci, _vp, _vp]),
    "btr_points_in_boxes": (_ci, [_ci] * 5 + [_vp] * 6),
    "btr_box3d_iou": (_ci, [_ci, _ci, _ci, _vp, _vp, _vp, _vp]),
}


def _load(mode=1):
    path = os.path.join(_PKG, "lib", _LIB_NAMES[mode])
    if not os.path.exists(path):
        raise ImportError(
            "Could not import _ext: %s is missing.\n"
            "Build the HIP extension first: python -c 'import __graft_entry__ as g; g.build()' "
            "(or python backtoreality_amd/build.py)." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = stale .so: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.btr_abi_version() != 1:
        raise ImportError("%s ABI %d != 1: rebuild" % (path, lib.btr_abi_version()))
    if lib.btr_distance_mode() != mode:
        raise ImportError("%s was built with BTR_FMAD=%d" % (path, lib.btr_distance_mode()))
    return lib


# `_lib`: the library behind everything that does not depend on the distance rounding (fused
# MLP, loss, evaluation kernels; callers bind it once).  `_idx`: the library the
# index-producing ops below call -- the same object unless BTR_FMAD / set_fmad() says otherwise.
_lib = _load(1)
_FMAD = int(os.environ.get("BTR_FMAD", "1"))
_idx = _lib if _FMAD == 1 else _load(_FMAD)


def fmad():
    """Rounding mode of the squared distance the index-producing ops currently use."""
    return _FMAD


def set_fmad(mode):
    """Select the library the index-producing ops (FPS, ball query, three_nn,
    three_interpolate) run from: 1 (default) / 0 / 2, see btr_distance_mode in the header."""
    global _FMAD, _idx
    mode = int(mode)
    _idx = _lib if mode == 1 else _load(mode)
    _FMAD = mode


class CompactRows(ctypes.Structure):
    """btr_compact_t of include/btr_pointnet2.h: device pointers describing the compact rows of
    one set-abstraction call; keeps the tensors alive while bound."""
    _fields_ = [("dims", _vp), ("bw", _vp), ("bgrp", _vp), ("goff", _vp),
                ("dense_rows", ctypes.c_double)]


MAX_LAYERS = 8   # BTR_MAX_LAYERS
_vp8, _ci8, _cf8, _sz8 = _vp * 8, _ci * 8, _cf * 8, _sz * 8
SA_OPT_COMPACT, SA_OPT_RECOMPUTE, SA_OPT_POOL_EPILOGUE, SA_OPT_POOL_GRAD = 1, 2, 4, 8
SA_OPT_POOL_GRAM = 16
SA_OPT_PPFL = 64
SA_OPT_PPFL_XYZ = 128
SA_OPT_EVAL = 256


class SaLayer(ctypes.Structure):
    """btr_sa_layer_t: one set-abstraction layer (sizes, parameter pointers, options)."""
    _fields_ = [("b", _ci), ("n", _ci), ("m", _ci), ("s", _ci), ("c", _ci), ("use_xyz", _ci),
                ("radius_div", _cf), ("layers", _ci), ("width", _ci8), ("w", _vp8),
                ("gamma", _vp8), ("beta", _vp8), ("running_mean", _vp8), ("running_var", _vp8),
                ("num_batches_tracked", _vp8), ("eps", _cf8), ("momentum", _cf8),
                ("need_dxyz", _ci), ("need_dnew_xyz", _ci), ("need_dfeat", _ci),
                ("options", _ci)]


class SaPlan(ctypes.Structure):
    """btr_sa_plan_t: the variant that runs + the offsets of every tensor in the buffers."""
    _fields_ = [("compact", _ci), ("recompute", _ci), ("pool_epilogue", _ci), ("pool_grad", _ci),
                ("rows", _ci), ("k0", _ci), ("k0p", _ci), ("kin", _ci8),
                ("x0", _sz), ("y", _sz8), ("w2", _sz8), ("wt", _sz8), ("stats", _sz8),
                ("arg", _sz), ("goff", _sz), ("dims", _sz), ("cidx", _sz), ("bgrp", _sz),
                ("bw", _sz), ("saved_bytes", _sz), ("fwd_scratch_bytes", _sz),
                ("bwd_scratch_bytes", _sz), ("dw", _sz8), ("dgamma", _sz8), ("dbeta", _sz8),
                ("grads_floats", _sz)]


class PmChain(ctypes.Structure):
    """btr_pm_chain_t: one conv1x1 (+bias) -> BatchNorm -> ReLU chain."""
    _fields_ = [("b", _ci), ("n", _ci), ("c", _ci), ("layers", _ci), ("width", _ci8),
                ("has_bn", _ci8), ("w", _vp8), ("bias", _vp8), ("gamma", _vp8), ("beta", _vp8),
                ("running_mean", _vp8), ("running_var", _vp8), ("num_batches_tracked", _vp8),
                ("eps", _cf8), ("momentum", _cf8), ("need_dx", _ci)]


class PmPlan(ctypes.Structure):
    _fields_ = [("rows", _ci), ("np", _ci8), ("kin", _ci8), ("x0", _sz), ("y", _sz8),
                ("w2", _sz8), ("wt", _sz8), ("stats", _sz8), ("saved_bytes", _sz),
                ("fwd_scratch_bytes", _sz), ("bwd_scratch_bytes", _sz), ("dw", _sz8),
                ("dbias", _sz8), ("dgamma", _sz8), ("dbeta", _sz8), ("grads_floats", _sz)]


MAX_LEVELS = 4   # BTR_MAX_LEVELS
_sz4, _ci4, _cf4 = _sz * 4, _ci * 4, _cf * 4


class Backbone(ctypes.Structure):
    """btr_backbone_t: L set-abstraction levels + F feature-propagation modules."""
    _fields_ = [("b", _ci), ("n", _ci), ("c", _ci), ("levels", _ci), ("fps", _ci),
                ("radius", _cf4), ("sa", SaLayer * 4), ("fp", PmChain * 4)]


class BackbonePlan(ctypes.Structure):
    _fields_ = [("sa", SaPlan * 4), ("fp", PmPlan * 4), ("g_xyz", _sz), ("g_feat", _sz),
                ("g_inds", _sz4), ("g_new_xyz", _sz4), ("g_idx", _sz4), ("g_fps_ws", _sz4),
                ("g_fps_ws_bytes", _sz4), ("g_fps_temp", _sz4), ("bq_buckets", _ci4),
                ("g_nn_idx", _sz4), ("g_nn_w", _sz4), ("g_goff", _sz4), ("g_dims", _sz4),
                ("g_cidx", _sz4), ("g_bgrp", _sz4), ("g_bw", _sz4), ("g_len", _sz4),
                ("g_scat", _sz4), ("g_scat_bytes", _sz4), ("g_ti", _sz4), ("g_ti_bytes", _sz4),
                ("g_ws", _sz), ("g_ws_bytes", _sz),
                ("geom_bytes", _sz), ("o_sa", _sz4), ("o_sa_cl", _sz4), ("o_fp", _sz4),
                ("o_fp_cl", _sz4), ("out_bytes", _sz), ("s_sa", _sz4), ("s_fp", _sz4),
                ("s_fpx", _sz4), ("saved_bytes", _sz), ("fwd_scratch_bytes", _sz),
                ("bwd_scratch_bytes", _sz), ("gr_sa", _sz4), ("gr_fp", _sz4),
                ("grads_floats", _sz)]


_vp3, _cf3, _sz3 = _vp * 3, _cf * 3, _sz * 3


class DecoderLayer(ctypes.Structure):
    """btr_decoder_layer_t: one GroupFree3D TransformerDecoderLayer."""
    _fields_ = [("b", _ci), ("pq", _ci), ("pk", _ci), ("e", _ci), ("heads", _ci), ("ff", _ci),
                ("dropout", _cf), ("seed", ctypes.c_ulonglong), ("step", _vp),
                ("sa_in_w", _vp), ("sa_in_b", _vp), ("sa_out_w", _vp), ("sa_out_b", _vp),
                ("ca_in_w", _vp), ("ca_in_b", _vp), ("ca_out_w", _vp), ("ca_out_b", _vp),
                ("lin1_w", _vp), ("lin1_b", _vp), ("lin2_w", _vp), ("lin2_b", _vp),
                ("ln_w", _vp3), ("ln_b", _vp3), ("ln_eps", _cf3)]


class DecoderPlan(ctypes.Structure):
    _fields_ = [("rq", _ci), ("rk", _ci)] + [(n, _sz) for n in (
        "qp0", "qkv", "a1", "lse1", "xh1", "rs1", "x1", "qp1", "q2", "kp", "kv", "a2", "lse2",
        "xh2", "rs2", "x2", "h", "xh3", "rs3", "saved_bytes", "fwd_scratch_bytes",
        "bwd_scratch_bytes", "g_sa_in_w", "g_sa_in_b", "g_sa_out_w", "g_sa_out_b", "g_ca_in_w",
        "g_ca_in_b", "g_ca_out_w", "g_ca_out_b", "g_lin1_w", "g_lin1_b", "g_lin2_w",
        "g_lin2_b")] + [("g_ln", _sz3), ("grads_floats", _sz)]


GF_MAX_DECODER_LAYERS = 12   # BTR_GF_MAX_DECODER_LAYERS
_szd = _sz * GF_MAX_DECODER_LAYERS


class GfStack(ctypes.Structure):
    """btr_gf_stack_t: the decoder layers with their position embeddings and prediction heads."""
    _fields_ = [("layers", _ci), ("b", _ci), ("pq", _ci), ("pk", _ci), ("e", _ci), ("nh", _ci),
                ("ns", _ci), ("head_c", _ci), ("has_qpos", _ci), ("has_kpos", _ci),
                ("layer", DecoderLayer * GF_MAX_DECODER_LAYERS),
                ("qpos", PmChain * GF_MAX_DECODER_LAYERS),
                ("kpos", PmChain * GF_MAX_DECODER_LAYERS),
                ("head", PmChain * GF_MAX_DECODER_LAYERS)]


class GfStackPlan(ctypes.Structure):
    _fields_ = [("layer", DecoderPlan * GF_MAX_DECODER_LAYERS),
                ("qpos", PmPlan * GF_MAX_DECODER_LAYERS),
                ("kpos", PmPlan * GF_MAX_DECODER_LAYERS),
                ("head", PmPlan * GF_MAX_DECODER_LAYERS),
                ("s_layer", _szd), ("s_qpos", _szd), ("s_kpos", _szd), ("s_head", _szd),
                ("s_x", _szd), ("s_qpos_cl", _szd), ("s_kpos_cl", _szd),
                ("saved_bytes", _sz), ("fwd_scratch_bytes", _sz), ("bwd_scratch_bytes", _sz),
                ("g_layer", _szd), ("g_qpos", _szd), ("g_kpos", _szd), ("g_head", _szd),
                ("grads_floats", _sz)]


class AdamItem(ctypes.Structure):
    """btr_adam_item_t"""
    _fields_ = [("p", _vp), ("m", _vp), ("v", _vp), ("step", _vp), ("n", ctypes.c_longlong),
                ("group", _ci), ("vec", _ci)]


ADAM_MAX_TENSORS = 448   # BTR_ADAM_MAX_TENSORS
ADAM_MAX_GROUPS = 8      # BTR_ADAM_MAX_GROUPS


class AdamGroups(ctypes.Structure):
    """btr_adam_groups_t"""
    _fields_ = [("lr", _cf * ADAM_MAX_GROUPS), ("wd", _cf * ADAM_MAX_GROUPS)]


class AdamGrads(ctypes.Structure):
    """btr_adam_grads_t"""
    _fields_ = [("g", _vp * ADAM_MAX_TENSORS)]


class GfLoss(ctypes.Structure):
    """btr_gf_loss_t"""
    _fields_ = [(n, _ci) for n in ("b", "p", "k2", "nh", "ns", "nc", "heads", "c", "s1", "n")] + \
               [(n, _cf) for n in ("w_obj", "w_box", "w_sem", "center_delta", "heading_delta",
                                   "size_delta")]


class compact_bound(object):
    """`with compact_bound(cm):` -- the btr_sa_* calls inside operate on compact rows
    (btr_sac_bind is per host thread; autograd's backward thread binds its own)."""

    def __init__(self, cm):
        self.cm = cm

    def __enter__(self):
        if self.cm is not None:
            _lib.btr_sac_bind(ctypes.addressof(self.cm))
        return self.cm

    def __exit__(self, *exc):
        if self.cm is not None:
            _lib.btr_sac_bind(None)
        return False


if __package__:
    from ._twin import attach_derived, attach_twin, derived, twin_of  # noqa: E402,F401
else:
    from _twin import attach_derived, attach_twin, derived, twin_of  # noqa: E402,F401


# ------------------------------------------------------------------------------------ checks
def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check(t, name, kind, like=None):
    _require(isinstance(t, torch.Tensor), "%s must be a tensor" % name)
    _require(t.is_contiguous(), "%s must be a contiguous tensor" % name)
    if kind == "float":
        _require(t.dtype == torch.float32, "%s must be a float tensor" % name)
    else:
        _require(t.dtype == torch.int32, "%s must be an int tensor" % name)
    if like is not None and like.is_cuda:
        _require(t.is_cuda, "%s must be a CUDA tensor" % name)
        _require(t.device == like.device, "%s must be on %s" % (name, like.device))


def _gpu_only(t):
    _require(t.is_cuda, "CPU not supported")


def _p(t):
    return t.data_ptr() if t is not None else None


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


_TIMING = None  # None = off; else list of (op, shape-key, start_event, end_event)
_TIMING_FILTER = None


def timing_begin(select=None):
    """Start recording a HIP event pair around kernel launches made through this module (on
    the launch stream).  `select(op, key) -> bool` restricts which launches are bracketed (an
    event pair costs a few microseconds of host time).  Used by bench.py for the live
    per-kernel durations."""
    global _TIMING, _TIMING_FILTER
    _TIMING = []
    _TIMING_FILTER = select


PAIR_OVERHEAD_MS = 0.0   # what timing_end() last measured for an EMPTY event pair


def timing_end():
    """Stop recording; returns {(op, shape-key): [ms per launch, ...]} (synchronises).
    An event pair recorded by `_call` around a launch reads the kernel's duration PLUS the
    distance two back-to-back markers have on the stream (a few microseconds: as much as the
    shortest kernels take).  That distance is measured here -- the lower quartile of 32 empty pairs
    queued behind a backlog on the same stream -- and subtracted from those entries (never below zero); pairs the library
    records itself right around one kernel (fps_kernel, ball query) are left as they are."""
    global _TIMING, PAIR_OVERHEAD_MS
    rec, _TIMING = _TIMING or [], None
    empty = []
    if rec and torch.cuda.is_available():
        # behind a backlog (a ~0.5 ms spin kernel), so that the markers' distance on the stream is
        # the GPU's and not the host's pace of issuing them: on a busy host two record() calls
        # were 37 us apart and the "overhead" wiped out the kernels it was subtracted from
        torch.cuda._sleep(1000000)
        for _ in range(32):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            b.record()
            empty.append((a, b))
    torch.cuda.synchronize()
    gaps = sorted(a.elapsed_time(b) for a, b in empty)
    PAIR_OVERHEAD_MS = gaps[len(gaps) // 4] if gaps else 0.0   # lower quartile
    out = {}
    for item in rec:
        op, key, e0, e1 = item[:4]
        t = e0.elapsed_time(e1)
        if len(item) > 4 and item[4]:
            t = max(t - PAIR_OVERHEAD_MS, 0.0)
        out.setdefault((op, key), []).append(t)
    return out


def timing_detail():
    """True inside a fully instrumented region (timing_begin() without a filter)."""
    return _TIMING is not None and _TIMING_FILTER is None


def _call(fn, *args, key=None):
    timed = _TIMING is not None and key is not None
    if timed and _TIMING_FILTER is not None:
        timed = _TIMING_FILTER(fn.__name__.replace("btr_", "").replace("_ws", ""), key)
    if timed:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = fn(*args)
    if timed:
        e1.record()
        _TIMING.append((fn.__name__.replace("btr_", "").replace("_ws", ""), key, e0, e1, True))
    if rc != 0:   # the error text is thread-local inside the library that owns `fn`
        owner = _idx if getattr(_idx, fn.__name__, None) is fn else _lib
        raise RuntimeError("%s failed (%d): %s" %
                           (fn.__name__, rc, owner.btr_last_error().decode(errors="replace")))


class _on(object):
    """Device guard: the reference has none (it relies on the caller's current device)."""

    def __init__(self, t):
        self.idx = t.device.index
        self.guard = None

    def __enter__(self):
        if self.idx != torch.cuda.current_device():
            self.guard = torch.cuda.device(self.idx)
            self.guard.__enter__()
        return self.idx

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


# -------------------------------------------------------------------------- the nine callables
def furthest_point_sampling(points, nsamples):
    """(B,N,3) f32 -> (B,nsamples) i32.  sampling.cpp:70-91."""
    _check(points, "points", "float")
    _gpu_only(points)
    _require(points.dim() == 3 and points.size(2) == 3, "points must be (B, N, 3)")
    B, N, _ = points.shape
    nsamples = int(nsamples)
    out = torch.empty((B, max(nsamples, 0)), dtype=torch.int32, device=points.device)
    if nsamples <= 0 or B == 0:
        return out
    return _fps(points, nsamples, 0, out)


def graph_stats():
    """{replays, captures, eager}: calls of the graph-capable entry points (the GroupFree3D decoder
    stack) replayed from a captured HIP graph / captured / issued launch by launch since load."""
    v = (ctypes.c_longlong * 3)()
    base = ctypes.addressof(v)
    _lib.btr_graph_stats(base, base + 8, base + 16)
    return {"replays": int(v[0]), "captures": int(v[1]), "eager": int(v[2])}


# Bumped by every library call that writes BatchNorm running statistics or parameters through raw
# pointers (training forwards, graph replays, the native Adam step): tensor version counters do
# not see those writes, so constants derived from them for the inference forward
# (fused_sa._eval_constants, fused_mlp._eval_chain_constants) carry this in their cache key.
RUNNING_STATS_EPOCH = [0]


def new_stream(device):
    """A new torch stream for the library's side work (the next batch's sampling pyramid)."""
    return torch.cuda.Stream(device=device)


def set_fps_lds_kb(kb):
    """Process-wide override of the dynamic LDS the large-scene FPS launch asks for (its running
    min-dists live there; 0: all of them in global memory; < 0: the default rules,
    include/btr_pointnet2.h btr_fps_lds_kb).  Never changes a result."""
    _idx.btr_fps_set_lds_kb(int(kb))   # (the library the index-producing ops run from: set_fmad)


def mark_fps_ordered(points):
    """`points` (B, M, 3) are the points an FPS sampled, in sampling order (new_xyz of a
    set-abstraction layer).  A later FPS of that tensor -- the next level of the pyramid --
    then first checks in parallel whether its answer is 0, 1, 2, ... (csrc/sampling.hip
    fps_prefix_check_kernel).  A hint about speed only: the indices are those of the plain call
    whatever the tensor holds (tests/test_ops_gpu.py), and it lapses when the tensor is
    written to."""
    points._btr_fps_ordered = points._version
    return points


def _fps(points, nsamples, block_size, out):
    B, N, _ = points.shape
    if getattr(points, "_btr_fps_ordered", None) == points._version:
        need = _idx.btr_fps_ordered_scratch_bytes(B, N, nsamples)
        if need:
            scratch = torch.empty((need,), dtype=torch.uint8, device=points.device)
            with _on(points) as dev:
                _call(_idx.btr_furthest_point_sampling_ordered, B, N, nsamples, _p(points), None,
                      _p(out), int(block_size), _p(scratch), need, _stream(dev),
                      key=(B, N, nsamples))
            return out
    ws_bytes = _idx.btr_furthest_point_sampling_workspace_bytes(B, N, nsamples)
    if ws_bytes:   # bucketed kernel: scratch from torch's stream-ordered caching allocator
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=points.device)
        temp = None
    else:          # register-resident kernel (no scratch) or streaming kernel (uses temp)
        ws = None
        temp = torch.empty((B, N), dtype=torch.float32, device=points.device)
    with _on(points) as dev:
        kernel_only = ws is not None and _TIMING is not None and (
            _TIMING_FILTER is None or _TIMING_FILTER("fps_kernel", (B, N, nsamples)))
        if kernel_only:   # event pair around the sampling kernel alone (not its sort launches)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()   # (creates the handles; the library records them again in place)
            e1.record()
            _idx.btr_fps_time_next_kernel(e0.cuda_event, e1.cuda_event)
            _TIMING.append(("fps_kernel", (B, N, nsamples), e0, e1))
        _call(_idx.btr_furthest_point_sampling_ws, B, N, nsamples, _p(points), _p(temp),
              _p(out), int(block_size), _p(ws), ws_bytes, _stream(dev),
              key=(B, N, nsamples))
        if ws is not None:
            # the cloud, sorted into Hilbert buckets, stays with the tensor it was built from:
            # a ball query of the SAME tensor (SA1 right after its FPS) searches it instead of
            # building a grid of its own (ball_query below; csrc/ball_query_bucket.hip)
            points._btr_fps_ws = (ws, points._version, torch.cuda.current_stream(dev), _idx)
    return out


def furthest_point_sampling_bs(points, nsamples, block_size):
    """Test hook: FPS with the reference block size forced (tie-break per instantiation)."""
    _check(points, "points", "float")
    _gpu_only(points)
    B, N, _ = points.shape
    out = torch.empty((B, max(int(nsamples), 0)), dtype=torch.int32, device=points.device)
    _require(int(block_size) >= 1, "block_size must be >= 1")
    if int(nsamples) <= 0 or B == 0:
        return out
    return _fps(points, int(nsamples), int(block_size), out)


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M).  sampling.cpp:20-44."""
    _check(points, "points", "float")
    _check(idx, "idx", "int", like=points)
    _gpu_only(points)
    B, C, N = points.shape
    M = idx.size(1)
    out = torch.empty((B, C, M), dtype=torch.float32, device=points.device)
    with _on(points) as dev:
        _call(_lib.btr_gather_points, B, C, N, M, _p(points), _p(idx), _p(out), _stream(dev))
    return out


def gather_rows(src, idx):
    """(B,N,C) f32, (B,M) i32 -> (B,M,C): the rows `idx` of a channel-last tensor."""
    _check(src, "src", "float")
    _check(idx, "idx", "int", like=src)
    _gpu_only(src)
    _require(src.dim() == 3 and idx.dim() == 2 and idx.size(0) == src.size(0),
             "src must be (B, N, C) and idx (B, M)")
    B, N, C = src.shape
    M = idx.size(1)
    out = torch.empty((B, M, C), dtype=torch.float32, device=src.device)
    with _on(src) as dev:
        _call(_lib.btr_gather_rows, B, N, M, C, _p(src), _p(idx), _p(out), _stream(dev))
    return out


def gather_rows_grad(grad_out, idx, n):
    """(B,M,C) f32, (B,M) i32, n -> (B,n,C): gradient of gather_rows, one launch."""
    _check(grad_out, "grad_out", "float")
    _check(idx, "idx", "int", like=grad_out)
    _gpu_only(grad_out)
    _require(grad_out.dim() == 3 and idx.dim() == 2 and idx.shape == grad_out.shape[:2],
             "grad_out must be (B, M, C) and idx (B, M)")
    B, M, C = grad_out.shape
    out = torch.empty((B, n, C), dtype=torch.float32, device=grad_out.device)
    with _on(grad_out) as dev:
        _call(_lib.btr_gather_rows_grad, B, n, M, C, _p(grad_out), _p(idx), _p(out), _stream(dev))
    return out


def gather_points_grad(grad_out, idx, n):
    """(B,C,M) f32, (B,M) i32, n -> (B,C,n).  sampling.cpp:46-69."""
    _check(grad_out, "grad_out", "float")
    _check(idx, "idx", "int", like=grad_out)
    _gpu_only(grad_out)
    B, C, M = grad_out.shape
    out = torch.empty((B, C, int(n)), dtype=torch.float32, device=grad_out.device)
    with _on(grad_out) as dev:
        _call(_lib.btr_gather_points_grad, B, C, int(n), M, _p(grad_out), _p(idx), _p(out),
              _stream(dev))
    return out


BQ_CALLS = {"buckets": 0, "own": 0}   # which path ball_query took (tests / tools read it)


def ball_query(new_xyz, xyz, radius, nsample):
    """(B,M,3), (B,N,3) f32 -> (B,M,nsample) i32.  C++ argument order, ball_query.cpp:13-37."""
    _check(new_xyz, "new_xyz", "float")
    _check(xyz, "xyz", "float", like=new_xyz)
    _gpu_only(new_xyz)
    B, M, _ = new_xyz.shape
    N = xyz.size(1)
    nsample = int(nsample)
    out = torch.empty((B, M, nsample), dtype=torch.int32, device=new_xyz.device)
    shared = getattr(xyz, "_btr_fps_ws", None)
    # (not while a HIP graph is being captured: the tensor identity / version test that makes
    # the reuse safe is evaluated once at capture time, the replays would keep reading a
    # workspace that belongs to another batch)
    if (shared is not None and shared[1] == xyz._version and shared[3] is _idx and
            os.environ.get("BTR_BQ_BUCKETS", "1") != "0" and
            not torch.cuda.is_current_stream_capturing()):
        bws = _idx.btr_ball_query_buckets_workspace_bytes(B, N, M, nsample)
        if bws:
            fps_ws, _, fps_stream, _ = shared
            with _on(new_xyz) as dev:
                cur = torch.cuda.current_stream(dev)
                if cur != fps_stream:        # built on another stream (prefetched pyramid):
                    fps_ws.record_stream(cur)  # the caller has already waited for that FPS
                box = torch.empty((bws,), dtype=torch.uint8, device=new_xyz.device)
                _call(_idx.btr_ball_query_buckets, B, N, M, float(radius), nsample, _p(new_xyz),
                      _p(fps_ws), _p(out), _p(box), bws, _stream(dev), key=(B, N, M, nsample))
            BQ_CALLS["buckets"] += 1
            return out
    BQ_CALLS["own"] += 1
    ws_bytes = _idx.btr_ball_query_workspace_bytes(B, N, M, nsample)
    ws = (torch.empty((ws_bytes,), dtype=torch.uint8, device=new_xyz.device)
          if ws_bytes else None)
    with _on(new_xyz) as dev:
        _call(_idx.btr_ball_query_ws, B, N, M, float(radius), nsample, _p(new_xyz), _p(xyz),
              _p(out), _p(ws), ws_bytes, _stream(dev), key=(B, N, M, nsample))
    return out


def group_points(points, idx):
    """(B,C,N) f32, (B,M,S) i32 -> (B,C,M,S).  group_points.cpp:17-40."""
    _check(points, "points", "float")
    _check(idx, "idx", "int", like=points)
    _gpu_only(points)
    B, C, N = points.shape
    _, M, S = idx.shape
    out = torch.empty((B, C, M, S), dtype=torch.float32, device=points.device)
    with _on(points) as dev:
        _call(_lib.btr_group_points, B, C, N, M, S, _p(points), _p(idx), _p(out), _stream(dev),
              key=(B, C, N, M, S))
    return out


def group_points_grad(grad_out, idx, n):
    """(B,C,M,S) f32, (B,M,S) i32, n -> (B,C,n).  group_points.cpp:42-65."""
    _check(grad_out, "grad_out", "float")
    _check(idx, "idx", "int", like=grad_out)
    _gpu_only(grad_out)
    B, C, M, S = grad_out.shape
    out = torch.empty((B, C, int(n)), dtype=torch.float32, device=grad_out.device)
    with _on(grad_out) as dev:
        _call(_lib.btr_group_points_grad, B, C, int(n), M, S, _p(grad_out), _p(idx), _p(out),
              _stream(dev), key=(B, C, int(n), M, S))
    return out


def three_nn(unknowns, knows):
    """(B,n,3), (B,m,3) -> [dist2 (B,n,3) f32 (squared), idx (B,n,3) i32].  interpolate.cpp:19-45."""
    _check(unknowns, "unknowns", "float")
    _check(knows, "knows", "float", like=unknowns)
    _gpu_only(unknowns)
    B, n, _ = unknowns.shape
    m = knows.size(1)
    dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknowns.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknowns.device)
    with _on(unknowns) as dev:
        _call(_idx.btr_three_nn, B, n, m, _p(unknowns), _p(knows), _p(dist2), _p(idx),
              _stream(dev))
    return [dist2, idx]


def three_nn_weights(unknowns, knows):
    """three_nn + the FP module's normalised inverse-distance weights in one launch:
    -> [dist2 (B,n,3), idx (B,n,3) i32, weight (B,n,3)]."""
    _check(unknowns, "unknowns", "float")
    _check(knows, "knows", "float", like=unknowns)
    _gpu_only(unknowns)
    B, n, _ = unknowns.shape
    m = knows.size(1)
    dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknowns.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknowns.device)
    weight = torch.empty((B, n, 3), dtype=torch.float32, device=unknowns.device)
    with _on(unknowns) as dev:
        _call(_idx.btr_three_nn_weights, B, n, m, _p(unknowns), _p(knows), _p(dist2), _p(idx),
              _p(weight), _stream(dev))
    return [dist2, idx, weight]


def three_interpolate(points, idx, weight):
    """(B,C,m) f32, (B,n,3) i32, (B,n,3) f32 -> (B,C,n).  interpolate.cpp:47-75."""
    _check(points, "points", "float")
    _check(idx, "idx", "int", like=points)
    _check(weight, "weight", "float", like=points)
    _gpu_only(points)
    B, C, m = points.shape
    n = idx.size(1)
    out = torch.empty((B, C, n), dtype=torch.float32, device=points.device)
    with _on(points) as dev:
        _call(_idx.btr_three_interpolate, B, C, m, n, _p(points), _p(idx), _p(weight), _p(out),
              _stream(dev))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """(B,C,n) f32, idx, weight, m -> (B,C,m).  interpolate.cpp:76-104."""
    _check(grad_out, "grad_out", "float")
    _check(idx, "idx", "int", like=grad_out)
    _check(weight, "weight", "float", like=grad_out)
    _gpu_only(grad_out)
    B, C, n = grad_out.shape
    out = torch.empty((B, C, int(m)), dtype=torch.float32, device=grad_out.device)
    with _on(grad_out) as dev:
        _call(_lib.btr_three_interpolate_grad, B, C, n, int(m), _p(grad_out), _p(idx),
              _p(weight), _p(out), _stream(dev))
    return out


def _check64(t, name, like):
    _require(isinstance(t, torch.Tensor) and t.is_contiguous() and t.dtype == torch.float64,
             "%s must be a contiguous double tensor" % name)
    _require(t.is_cuda and t.device == like.device, "%s must be on %s" % (name, like.device))


def nms_boxes(boxes, score, threshold, old_type=False, cls=None, valid=None):
    """Greedy NMS per scene (utils/nms.py:42-156).  boxes (B,K,4|6) f64 [mins, maxs], score
    (B,K) f64, cls (B,K) i32 or None, valid (B,K) u8 or None -> pick (B,K) u8."""
    _require(isinstance(boxes, torch.Tensor), "boxes must be a tensor")
    _gpu_only(boxes)
    _check64(boxes, "boxes", boxes)
    _check64(score, "score", boxes)
    _require(boxes.dim() == 3 and boxes.size(2) in (4, 6), "boxes must be (B, K, 4) or (B, K, 6)")
    B, K, D = boxes.shape
    _require(tuple(score.shape) == (B, K), "score must be (B, K)")
    if cls is not None:
        _check(cls, "cls", "int", like=boxes)
        _require(tuple(cls.shape) == (B, K), "cls must be (B, K)")
    if valid is not None:
        _require(valid.dtype == torch.uint8 and valid.is_contiguous() and valid.is_cuda and
                 tuple(valid.shape) == (B, K), "valid must be a contiguous (B, K) uint8 tensor")
    pick = torch.empty((B, K), dtype=torch.uint8, device=boxes.device)
    with _on(boxes) as dev:
        _call(_lib.btr_nms_boxes, B, K, D // 2, _p(boxes), _p(score), _p(cls), _p(valid),
              float(threshold), int(bool(old_type)), _p(pick), _stream(dev))
    return pick


def points_in_boxes(points, center, size, angle, cap):
    """count (B,K) i32 = min(cap, points of scene b inside box k); points (B,N,C>=3) f32 in
    upright-depth coordinates, boxes as get_3d_box takes them (camera coordinates), f64."""
    _check(points, "points", "float")
    _gpu_only(points)
    _require(points.dim() == 3 and points.size(2) >= 3, "points must be (B, N, >=3)")
    for t, name in ((center, "center"), (size, "size"), (angle, "angle")):
        _check64(t, name, points)
    B, N, C = points.shape
    K = angle.size(1)
    _require(tuple(center.shape) == (B, K, 3) and tuple(size.shape) == (B, K, 3) and
             tuple(angle.shape) == (B, K), "center/size must be (B, K, 3), angle (B, K)")
    count = torch.empty((B, K), dtype=torch.int32, device=points.device)
    with _on(points) as dev:
        _call(_lib.btr_points_in_boxes, B, N, K, C, int(cap), _p(points), _p(center), _p(size),
              _p(angle), _p(count), _stream(dev))
    return count


def box3d_iou(corners1, corners2):
    """(S,P,8,3) f64 x (S,G,8,3) f64 -> (S,P,G) f64: box3d_iou of utils/box_util.py:98-128."""
    _require(isinstance(corners1, torch.Tensor), "corners1 must be a tensor")
    _gpu_only(corners1)
    _check64(corners1, "corners1", corners1)
    _check64(corners2, "corners2", corners1)
    _require(corners1.dim() == 4 and tuple(corners1.shape[2:]) == (8, 3) and
             corners2.dim() == 4 and tuple(corners2.shape[2:]) == (8, 3) and
             corners1.size(0) == corners2.size(0), "corners must be (S, P, 8, 3) and (S, G, 8, 3)")
    S, P, G = corners1.size(0), corners1.size(1), corners2.size(1)
    iou = torch.empty((S, P, G), dtype=torch.float64, device=corners1.device)
    with _on(corners1) as dev:
        for s0 in range(0, S, 32768):
            s1 = min(S, s0 + 32768)
            _call(_lib.btr_box3d_iou, s1 - s0, P, G, _p(corners1[s0:s1]), _p(corners2[s0:s1]),
                  _p(iou[s0:s1]), _stream(dev))
    return iou


def build_id():
    """Digest of the sources the loaded library was built from (btr_build_id)."""
    return _lib.btr_build_id().decode()


def opt_n_threads(work_size):
    return int(_lib.btr_opt_n_threads(int(work_size)))
