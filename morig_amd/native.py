"""ctypes binding of libmorig_hip.so (the C ABI in include/morig_hip.h) + the op layer the
network plans are written against.

There is NO fallback: if the shared library is missing, or a tensor is not on a ROCm device, the
product path raises. (tests/ inject a torch emulation of the *same op interface* to check the host
logic on CPU; that emulation lives under tests/ and is never importable from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MORIG_HIP_LIB") or os.path.join(_HERE, "lib", "libmorig_hip.so")   # env: A/B builds

c_f32p = C.c_void_p
c_i32p = C.c_void_p
c_i64p = C.c_void_p
c_f64p = C.c_void_p
c_u8p = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("X", c_f32p), ("ldx", C.c_int32),
        ("W", c_f32p), ("ldw", C.c_int32),
        ("bias", c_f32p), ("scale", c_f32p), ("shift", c_f32p),
        ("relu", C.c_int32),
        ("rowbias", c_f32p), ("ld_rowbias", C.c_int32),
        ("seg", c_i32p),
        ("Y", c_f32p), ("ldy", C.c_int32),
        ("pool", c_f32p), ("ld_pool", C.c_int32), ("n_seg", C.c_int32),
        ("W_split", C.c_void_p), ("overflow", c_i32p),
        ("x_split", C.c_int32), ("y_split", C.c_int32), ("w_split_format", C.c_int32),
        ("X_tail", c_f32p), ("ld_tail", C.c_int32), ("tail_rows", C.c_int32), ("tail_cols", C.c_int32),
    ]


class EdgeConvArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("H", C.c_int32),
        ("n_nodes", C.c_int32), ("replicas", C.c_int32),
        ("in_rep_stride", C.c_int32), ("out_rep_stride", C.c_int32),
        ("A", c_f32p), ("lda", C.c_int32),
        ("B", c_f32p), ("ldb", C.c_int32),
        ("rowptr", c_i32p), ("src_sorted", c_i32p), ("dst_sorted", c_i32p),
        ("edge_capacity", C.c_int32), ("edge_count", C.c_int32),
        ("s1", c_f32p), ("t1", c_f32p),
        ("W2", c_f32p), ("ldw", C.c_int32),
        ("b2", c_f32p), ("s2", c_f32p), ("t2", c_f32p),
        ("out", c_f32p), ("ldo", C.c_int32),
        ("W2_split", C.c_void_p), ("overflow", c_i32p),
        ("quad_aligned", C.c_int32),
        ("out_split", C.c_int32),
        ("exact_arith", C.c_int32),
        ("seg_min4", C.c_int32),
        ("init_with", C.c_void_p), ("split_with", C.c_void_p),          # the boundary passes of a pair of launches (include/morig_hip.h)
        ("skip_init", C.c_int32), ("skip_split", C.c_int32),
    ]


class EdgeConvX3Args(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("H", C.c_int32),
        ("n_nodes", C.c_int32), ("replicas", C.c_int32),
        ("in_rep_stride", C.c_int32), ("out_rep_stride", C.c_int32),
        ("X", c_f32p), ("ldx", C.c_int32),
        ("W1a", c_f32p), ("W1b", c_f32p), ("b1", c_f32p),
        ("rowptr", c_i32p), ("src_sorted", c_i32p), ("dst_sorted", c_i32p),
        ("edge_capacity", C.c_int32), ("edge_count", C.c_int32),
        ("W2", c_f32p), ("ldw", C.c_int32),
        ("b2", c_f32p), ("s2", c_f32p), ("t2", c_f32p),
        ("out", c_f32p), ("ldo", C.c_int32),
        ("W2_split", C.c_void_p), ("overflow", c_i32p),
        ("init_with", C.c_void_p), ("skip_init", C.c_int32), ("reserved0", C.c_int32),
    ]


class PointConvArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("A", c_f32p), ("lda", C.c_int32), ("B", c_f32p), ("ldb", C.c_int32),
        ("slots", C.c_void_p), ("max_nbrs", C.c_int32), ("n_centres", C.c_int32), ("n_src", C.c_int32),
        ("H", C.c_int32), ("H3", C.c_int32),
        ("W2_split", C.c_void_p), ("ldw2", C.c_int32), ("b2", c_f32p),
        ("W3_split", C.c_void_p), ("ldw3", C.c_int32), ("b3", c_f32p), ("s3", c_f32p), ("t3", c_f32p), ("relu3", C.c_int32),
        ("out", c_f32p), ("ldo", C.c_int32), ("overflow", C.c_void_p), ("status", C.c_void_p),
    ]


class SegmaxArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("N", C.c_int32), ("K", C.c_int32),
        ("X", c_f32p), ("ldx", C.c_int32),
        ("W", c_f32p), ("ldw", C.c_int32),
        ("bias", c_f32p), ("scale", c_f32p), ("shift", c_f32p), ("relu", C.c_int32),
        ("rowptr", c_i32p), ("dst_sorted", c_i32p), ("n_nodes", C.c_int32),
        ("edge_capacity", C.c_int32), ("edge_count", C.c_int32),
        ("out", c_f32p), ("ldo", C.c_int32),
        ("W_split", C.c_void_p), ("overflow", c_i32p),
    ]


def _args(cls):
    """a zeroed argument struct with its struct_size set (ABI 3: the library refuses a struct shorter than its version-3 layout and reads
    members past struct_size as zero, include/morig_hip.h)"""
    a = cls()
    a.struct_size = C.sizeof(cls)
    return a


_SIGNATURES = {
    "morig_abi_version": (C.c_int, []),
    "morig_strerror": (C.c_char_p, [C.c_int]),
    "morig_reserve_cus": (C.c_int, [C.c_int]),
    "morig_last_hip_error": (C.c_int, []),
    "morig_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "morig_csr_build": (C.c_int, [c_i64p, C.c_int64, C.c_int32, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_void_p]),
    "morig_csr_build_bipartite": (C.c_int, [c_i64p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_void_p]),
    "morig_csr_from_slots": (C.c_int, [c_i64p, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_void_p]),
    "morig_gemm": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "morig_edgeconv_x3": (C.c_int, [C.POINTER(EdgeConvX3Args), C.c_void_p]),
    "morig_edge_hidden": (C.c_int, [C.POINTER(EdgeConvArgs), C.c_void_p]),
    "morig_segmax_gemm": (C.c_int, [C.POINTER(SegmaxArgs), C.c_void_p]),
    "morig_pointconv_fused": (C.c_int, [C.POINTER(PointConvArgs), C.c_void_p]),
    "morig_fps": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_i32p, c_i32p, C.c_int32, C.c_int32, c_i32p, C.c_void_p]),
    "morig_ball_query": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, C.c_float, C.c_int32, c_i64p, C.c_void_p]),
    "morig_radius_sample": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_uint32,
                                      c_i64p, c_i32p, C.c_void_p]),
    "morig_csr_build_dual": (C.c_int, [c_i64p, C.c_int64, C.c_int32, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_int32, c_i32p,
                                       C.c_void_p]),
    "morig_geo_ball_graph": (C.c_int, [c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_uint32,
                                       c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_void_p]),
    "morig_geo_ball_graph_dist": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int32, C.c_uint32,
                                            c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_void_p]),
    "morig_geo_ball_fill": (C.c_int, [c_i32p, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_i64p, C.c_int64, C.c_void_p]),
    "morig_col_stats": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_i32p, C.c_int32, C.c_void_p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "morig_col_affine": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_i32p, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_void_p]),
    "morig_bn_finalize": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, C.c_float, c_f32p, c_f32p, c_i64p, c_f32p, c_f32p, c_f32p,
                                    C.c_int32, C.c_void_p]),
    "morig_edge_gather_relu": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, c_i32p, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32,
                                         c_f32p, C.c_int32, C.c_void_p, C.c_int64, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "morig_segmax_affine": (C.c_int, [c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_void_p]),
    "morig_bn_backward_stats": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32, c_i32p, C.c_int32, c_f32p, c_f32p, C.c_void_p,
                                          C.c_int64, c_f32p, c_f32p, C.c_void_p]),
    "morig_bn_relu_backward": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32, c_i32p, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p,
                                         c_f32p, c_f32p, C.c_int32, C.c_void_p, C.c_int64, c_f32p, C.c_void_p]),
    "morig_segmax_affine_arg": (C.c_int, [c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, c_i32p,
                                          C.c_int32, c_f32p, C.c_int32, C.c_void_p]),
    "morig_segmax_bn_backward_stats": (C.c_int, [c_f32p, C.c_int32, c_i32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32,
                                                 C.c_int32, c_f32p, c_f32p, C.c_void_p, C.c_int64, c_f32p, c_f32p, C.c_void_p]),
    "morig_segmax_bn_relu_backward": (C.c_int, [c_f32p, C.c_int32, c_i32p, C.c_int32, c_f32p, C.c_int32, c_i32p, C.c_int32, c_i32p,
                                                C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, c_f32p, C.c_int32,
                                                C.c_void_p, C.c_int64, c_f32p, C.c_void_p]),
    "morig_edge_bn_scatter_backward": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, c_i32p, c_i32p, c_i32p, C.c_int32, C.c_int32,
                                                 C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, c_f32p, C.c_int32,
                                                 c_f32p, C.c_int32, c_f32p, C.c_int32, c_i32p, c_i32p, C.c_void_p]),
    "morig_edge_scatter_backward": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p,
                                              C.c_int32, C.c_void_p]),
    "morig_edge_bn_sums_from_products": (C.c_int, [c_f32p, C.c_int32, c_f32p, c_f32p, C.c_int32, c_f32p, c_f32p, C.c_int32, C.c_int32,
                                                   c_f32p, c_f32p, C.c_void_p]),
    "morig_gemm_tn_workspace": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "morig_gemm_tn": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32, c_i32p, C.c_int32, C.c_int32, c_f32p, C.c_int64, c_f32p,
                                C.c_int32, C.c_void_p]),
    "morig_gemm_tn_shift": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, c_f32p, C.c_int64,
                                      c_f32p, C.c_int32, C.c_void_p]),
    "morig_knn_interpolate": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_i32p, c_f32p, C.c_int32, c_i32p,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_f32p, c_f32p, C.c_int32, C.c_void_p]),
    "morig_knn_search": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   c_i32p, c_f32p, C.c_void_p]),
    "morig_knn_apply": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_i32p, c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_void_p]),
    "morig_cosine_nn": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_f32p, C.c_void_p]),
    "morig_sigmoid_minmax": (C.c_int, [c_f32p, C.c_int32, c_i32p, C.c_int32, c_f32p, C.c_int32, C.c_void_p]),
    "morig_cosine_knn": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    c_f32p, C.c_int32, C.c_int32, c_i32p, C.c_void_p]),
    "morig_flow_vote": (C.c_int, [C.c_int32, c_i32p, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32,
                                   c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_void_p]),
    "morig_inside_check": (C.c_int, [c_f64p, C.c_int32, c_u8p, c_f64p, C.c_double, C.c_double, c_u8p, C.c_void_p]),
    "morig_knn_bandwidth": (C.c_int, [c_f64p, C.c_int32, C.c_int32, c_f64p, c_f64p, C.c_void_p]),
    "morig_meanshift": (C.c_int, [c_f64p, c_f32p, C.c_int32, c_f64p, C.c_int32, c_f64p, c_f64p, c_f64p, c_i32p, C.c_void_p]),
    "morig_nms_counts": (C.c_int, [c_f64p, C.c_int32, c_f64p, c_i32p, C.c_void_p]),
    "morig_knn_bandwidth_batched": (C.c_int, [c_f64p, c_i32p, C.c_int32, C.c_int32, C.c_int32, C.c_double, c_f64p, c_f64p, C.c_void_p]),
    "morig_meanshift_batched": (C.c_int, [c_f64p, c_f32p, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_f64p, C.c_int32, c_f64p, c_f64p, c_f64p,
                                          c_i32p, C.c_void_p]),
    "morig_morton_keys": (C.c_int, [c_f64p, c_i32p, C.c_int32, C.c_int32, c_i64p, C.c_void_p]),
    "morig_meanshift_sorted": (C.c_int, [c_f64p, c_f32p, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_f64p, C.c_int32, c_f64p, c_f64p, c_f64p,
                                         c_f64p, c_i32p, C.c_void_p]),
    "morig_nms_counts_batched": (C.c_int, [c_f64p, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_f64p, c_i32p, C.c_void_p]),
    "morig_nms_counts_sorted": (C.c_int, [c_f64p, c_i32p, C.c_int32, C.c_int32, C.c_int32, c_f64p, c_f64p, c_i32p, C.c_void_p]),
    "morig_nms_greedy_batched": (C.c_int, [c_f64p, c_f32p, c_i32p, C.c_int32, C.c_int32, c_f64p, c_i32p, C.c_double, C.c_float, c_u8p,
                                           C.c_void_p]),
    "morig_nms_greedy": (C.c_int, [c_f64p, c_f32p, C.c_int32, c_f64p, c_i32p, C.c_double, C.c_float, c_u8p, C.c_void_p]),
    "morig_gather_rows": (C.c_int, [c_f32p, C.c_int32, c_i32p, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_void_p]),
    "morig_edgeconv": (C.c_int, [C.POINTER(EdgeConvArgs), C.c_void_p]),
    "morig_edgeconv_can_split_out": (C.c_int, [C.POINTER(EdgeConvArgs)]),
    "morig_copy2d": (C.c_int, [c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "morig_copy2d_pad": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_i32p, C.c_void_p]),
    "morig_copy2d_pad_rep": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                        C.c_int32, c_i32p, C.c_void_p]),
    "morig_gather_cols": (C.c_int, [c_f32p, C.c_int32, c_i32p, C.c_int32, c_f32p, C.c_int32, C.c_int32, C.c_void_p]),
    "morig_pack_tails": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32,
                                   c_f32p, C.c_int64, c_i32p, C.c_void_p]),
    "morig_make_seg": (C.c_int, [c_i64p, C.c_int32, C.c_int32, C.c_int32, c_i32p, C.c_void_p]),
    "morig_rownorm": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int32, C.c_void_p]),
    "morig_cls_attention": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_void_p]),
    "morig_frame_reduce": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_void_p]),
    "morig_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "morig_rccl_comm_init": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "morig_rccl_comm_destroy": (C.c_int, [C.c_void_p]),
    "morig_rccl_last_error": (C.c_int, []),
    "morig_allgather_rows": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_void_p]),
    "morig_allgather_counts": (C.c_int, [C.c_void_p, c_i64p, c_i64p, C.c_void_p]),
    "morig_prof_enable": (C.c_int, [C.c_int]),
    "morig_prof_reset": (C.c_int, []),
    "morig_prof_name": (C.c_char_p, [C.c_int]),
    "morig_prof_symbol": (C.c_char_p, [C.c_int]),
    "morig_ubench_mfma": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "morig_prof_collect": (C.c_int, [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

EXPORTS = tuple(_SIGNATURES)
ABI_VERSION = 3                # include/morig_hip.h MORIG_ABI_VERSION
_lib = None


class MorigNativeError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH):
    """dlopen libmorig_hip.so and type every export. Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise MorigNativeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C morig_amd/csrc`). There is no CPU fallback for the MoRig forward path.")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the header and the build disagree
        fn.restype, fn.argtypes = res, args
    if lib.morig_abi_version() != ABI_VERSION:
        raise MorigNativeError(f"libmorig_hip.so ABI version {lib.morig_abi_version()}, this binding is written against {ABI_VERSION} "
                               "(argument structs start with struct_size since 3): rebuild with `make -C morig_amd/csrc`")
    _lib = lib
    return lib


def library_path() -> str:
    """the file load_library() maps (MORIG_HIP_LIB selects an A/B build)"""
    return LIB_PATH


def check(status: int, what: str) -> None:
    if status != 0:
        lib = load_library()
        msg = lib.morig_strerror(status).decode()
        if status == -3:
            msg += f" [hipError_t={lib.morig_last_hip_error()}]"
        raise MorigNativeError(f"{what}: {msg}")


PAIR_PASSES = os.environ.get("MORIG_EDGE_PAIR", "1") != "0"


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------------------------------------
# the op interface the plans use
# ---------------------------------------------------------------------------------------------
@dataclass
class Mat:
    """A column window of a row-major fp32 2-D tensor: rows [row0, row0+rows), cols [col0, col0+cols)."""
    base: torch.Tensor
    row0: int
    col0: int
    rows: int
    cols: int

    @staticmethod
    def of(t: torch.Tensor, col0: int = 0, cols: Optional[int] = None, row0: int = 0, rows: Optional[int] = None) -> "Mat":
        assert t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1, "need a row-major fp32 matrix"
        cols = t.shape[1] - col0 if cols is None else cols
        rows = t.shape[0] - row0 if rows is None else rows
        assert 0 <= col0 and col0 + cols <= t.shape[1] and 0 <= row0 and row0 + rows <= t.shape[0]
        return Mat(t, row0, col0, rows, cols)

    @property
    def ld(self) -> int:
        return self.base.stride(0)

    @property
    def ptr(self) -> int:
        return self.base.data_ptr() + 4 * (self.row0 * self.ld + self.col0)

    def view(self) -> torch.Tensor:
        return self.base[self.row0:self.row0 + self.rows, self.col0:self.col0 + self.cols]


@dataclass
class CSR:
    rowptr: torch.Tensor      # int32 [n+1]
    src: torch.Tensor         # int32 [cap]
    dst: torch.Tensor         # int32 [cap]
    n_nodes: int
    capacity: int
    status: torch.Tensor      # int32 [1], non-zero = index out of range (read once per forward by NativeOps.guarded)
    edge_count: int = 0       # exact E' when known (accounting only)
    quad: bool = False        # segments padded to multiples of 4 (MORIG_CSR_PAD4)
    min4: bool = False        # segments of at least 4 rows, not aligned (MORIG_CSR_MIN4: the mixed-quad form of the H = 256 EdgeConv kernel)
    _transposed: Optional[tuple] = None

    def transposed(self, n_src: Optional[int] = None):
        """-> (rowptr_t int32 [n_src + 1], perm_t int32 [capacity]): the same edges grouped by SOURCE, perm_t[k] = the row of this CSR
        holding the k-th edge in (source, row) order -- what ``edge_bn_scatter_backward`` walks to sum the gradient of the source
        side in a fixed order. Built once per graph with stream-ordered torch calls (stable sort; no host read: the live row count
        stays on the device, rows past it sort behind every vertex)."""
        n_src = self.n_nodes if n_src is None else n_src
        if self._transposed is None or self._transposed[0] != n_src:
            dev = self.src.device
            live = self.rowptr[self.n_nodes]
            pos = torch.arange(self.capacity, device=dev)
            keys = torch.where(pos < live, self.src.long().clamp(0, n_src), torch.full((), n_src, dtype=torch.int64, device=dev))
            skeys, order = torch.sort(keys, stable=True)
            rowptr_t = torch.searchsorted(skeys, torch.arange(n_src + 1, device=dev)).int()
            self._transposed = (n_src, rowptr_t.contiguous(), order.int().contiguous())
        return self._transposed[1], self._transposed[2]


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MorigNativeError("the MoRig native forward path needs ROCm device tensors (no CPU fallback); "
                                   "move the model and data to 'cuda'")


class PendingGuard:
    """What a forward enqueued with ``NativeOps.guarded_async`` leaves behind: ONE small device tensor [range flag, CSR status
    words ...] snapshotted on the stream behind the forward's last launch. ``result()`` is the host read the synchronous guard
    does at the end of every forward -- a serving loop calls it AFTER it has enqueued the next forward, so the GPU never idles
    on it. -> True: results valid; False: an operand left the split-fp16 range (re-run on the fp32 path); raises on a bad index."""

    def __init__(self, snapshot: torch.Tensor):
        self.snapshot = snapshot
        self._done = None

    def result(self) -> bool:
        if self._done is None:
            words = self.snapshot.tolist()
            if any(w != 0 for w in words[1:]):
                raise MorigNativeError("edge_index / neighbour index out of range for the vertex count it was built with "
                                       "(morig_csr_build status %s)" % [w for w in words[1:] if w != 0][:4])
            self._done = words[0] == 0
        return self._done


class NativeOps:
    """Thin, validating wrappers: tensors in, C ABI calls out, everything on the current stream."""

    name = "hip"

    def __init__(self):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise MorigNativeError("no ROCm device visible: the MoRig forward path has no CPU fallback")
        # arithmetic of the MFMA contractions: "f16x3" = split-fp16 fast path with overflow guard
        # (default), "f32" = fp32 MFMA everywhere. MORIG_PRECISION overrides.
        self.precision = os.environ.get("MORIG_PRECISION", "f16x3")
        assert self.precision in ("f16x3", "f32")
        # arithmetic of the EXACT path (precision "f32", the re-run behind the range guard, the train-mode forward): "bf16x6" = both fp32
        # operands split into three bf16 limbs in the kernel, six MFMAs per product (float32-class products, float32's range, 2.7x the
        # fp32-MFMA rate: MORIG_SPLIT_BF16X6); "f32" = v_mfma_f32_32x32x2_f32. MORIG_EXACT_ARITH overrides.
        self.exact_arith = os.environ.get("MORIG_EXACT_ARITH", "bf16x6")
        assert self.exact_arith in ("bf16x6", "f32")
        # guard state is PER THREAD (two threads / streams may run forwards concurrently on the one NativeOps): nesting depth,
        # forced-fp32 switch, the CSR status words of the forward in flight and the overflow flag word of each device
        self._tls = threading.local()
        # accounting only (bench.py): exact self-loop-normalised edge counts E' per input graph, learned during
        # warm-up with one host read each, so the live FLOP counters use ALGORITHMIC edges (no capacity, no padding)
        self.learn_edge_counts = False
        self._edge_counts = {}

    # -- split-fp16 guard -------------------------------------------------------------------------
    @property
    def fast(self) -> bool:
        return self.precision == "f16x3" and not self._force_f32

    @property
    def split_activations(self) -> bool:
        """plans keep GEMM->GEMM activations in the split-fp16 layout while the fast path is active"""
        return self.fast and os.environ.get("MORIG_SPLIT_ACT", "1") != "0"

    def _state(self):
        t = self._tls
        if not hasattr(t, "depth"):
            t.depth, t.force_f32, t.csr_status, t.ovf = 0, False, None, {}
        return t

    @property
    def _depth(self):
        return self._state().depth

    @_depth.setter
    def _depth(self, v):
        self._state().depth = v

    @property
    def _force_f32(self):
        return self._state().force_f32

    @_force_f32.setter
    def _force_f32(self, v):
        self._state().force_f32 = v

    @property
    def _csr_status(self):
        return self._state().csr_status

    @_csr_status.setter
    def _csr_status(self, v):
        self._state().csr_status = v

    def _flag(self, device) -> torch.Tensor:
        ovf = self._state().ovf
        key = (device.type, device.index)
        if key not in ovf:
            ovf[key] = torch.zeros(1, dtype=torch.int32, device=device)
        return ovf[key]

    def guarded(self, device, fn, rerun: bool = True, retry=None):
        """Run ``fn()`` (a whole forward) on the fast path; if any kernel reported an operand outside the
        fp16 range, run it again on the exact path. Nested calls run inside the outer guard.
        retry (optional callable -> bool): asked after a range overflow BEFORE the exact path is taken; True = the caller changed
        something (the rig networks raise their range shift: morig_amd.models.basic_modules.NativeModule.forward) and the fast
        path is worth another attempt, False = give up and run exactly.
        rerun=False (train-mode forward: it has side effects on the BatchNorm running buffers, so it must run ONCE):
        the whole forward runs on the exact fp32 MFMA path."""
        if self._depth > 0:
            return fn()
        if not rerun:
            old = self._force_f32
            self._force_f32 = True
            try:
                return self._run_once(device, fn)
            finally:
                self._force_f32 = old
        if not self.fast:
            return self._run_once(device, fn)          # exact-fp32 mode: no range flag to read, but the CSR status words still are
        flag = self._flag(device)
        while True:
            flag.zero_()
            self._depth += 1
            self._csr_status = []
            try:
                out = fn()
            finally:
                self._depth -= 1
                stats, self._csr_status = self._csr_status, None
            # ONE small D2H read per forward: the split-fp16 range flag and the status word of every CSR built on the way
            # (an out-of-range / negative edge or ball-query index is dropped by the count and fill kernels; the reference would
            # raise an index error, so does this)
            words = torch.cat([flag] + stats).tolist() if stats else [int(flag.item())]
            if any(w != 0 for w in words[1:]):
                raise MorigNativeError("edge_index / neighbour index out of range for the vertex count it was built with "
                                       "(morig_csr_build status %s)" % [w for w in words[1:] if w != 0][:4])
            if words[0] == 0 or retry is None or not retry():
                break
        if words[0] != 0:
            self._force_f32 = True
            try:
                out = fn()
            finally:
                self._force_f32 = False
        return out

    def guarded_async(self, device, fn):
        """``fn()`` (a whole eval forward) on the fast path WITHOUT the host read at its end -> (outputs, PendingGuard).
        The flag and the CSR status words of this forward are copied into a fresh tensor on the stream, so the next forward may
        be enqueued (it clears the shared flag word) before ``PendingGuard.result()`` is called. Also what a HIP-graph capture
        of a forward runs through (no host read inside a capture; every replay refreshes the snapshot)."""
        assert self._depth == 0, "guarded_async is for the outermost forward"
        flag = self._flag(device)
        flag.zero_()
        self._depth += 1
        self._csr_status = []
        try:
            out = fn()
        finally:
            self._depth -= 1
            stats, self._csr_status = self._csr_status, None
        return out, PendingGuard(torch.cat([flag] + stats))

    def _run_once(self, device, fn):
        """one pass with the CSR status words checked (no precision flag: the fp32 path cannot overflow fp16)"""
        self._depth += 1
        self._csr_status = []
        try:
            out = fn()
        finally:
            self._depth -= 1
            stats, self._csr_status = self._csr_status, None
        if stats and any(w != 0 for w in torch.cat(stats).tolist()):
            raise MorigNativeError("edge_index / neighbour index out of range for the vertex count it was built with")
        return out

    # -- allocation (PyTorch owns all device memory) -------------------------------------------
    def empty(self, rows, cols, device, dtype=torch.float32):
        return torch.empty((rows, cols), device=device, dtype=dtype)

    # -- graph --------------------------------------------------------------------------------
    def csr_build(self, edge_index: torch.Tensor, n_nodes: int, n_src: Optional[int] = None,
                  skip_negative: bool = False, pad4: bool = False, min4: bool = False) -> CSR:
        """pad4: pad every target's segment to a multiple of 4 by repeating its self loop (exact for
        max-aggregation); enables the in-register quad reduction of the EdgeConv epilogue.
        min4: fill every segment up to 4 rows instead (MORIG_CSR_MIN4: the mixed-quad form of the H = 256 kernel; not with pad4)."""
        assert not (pad4 and min4)
        _need_gpu(edge_index)
        ei = edge_index if (edge_index.dtype == torch.int64 and edge_index.is_contiguous()) else edge_index.long().contiguous()
        E = ei.shape[1]
        dev = ei.device
        cap = E + (4 if (pad4 or min4) else 1) * n_nodes
        rowptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
        src = torch.empty(cap, dtype=torch.int32, device=dev)
        dst = torch.empty(cap, dtype=torch.int32, device=dev)
        cursor = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
        status = torch.empty(1, dtype=torch.int32, device=dev)
        if n_src is None and not skip_negative and not pad4 and not min4:
            check(self.lib.morig_csr_build(_p(ei), E, n_nodes, _p(rowptr), _p(src), _p(dst), _p(cursor), _p(status), _stream()),
                  "morig_csr_build")
        else:
            check(self.lib.morig_csr_build_bipartite(_p(ei), E, n_nodes if n_src is None else n_src, n_nodes,
                                                     (1 if skip_negative else 0) | (2 if pad4 else 0) | (4 if min4 else 0), _p(rowptr), _p(src),
                                                     _p(dst), _p(cursor), _p(status), _stream()), "morig_csr_build_bipartite")
        csr = CSR(rowptr, src, dst, n_nodes, cap, status, quad=pad4, min4=min4)
        if getattr(self, "_csr_status", None) is not None:
            self._csr_status.append(status)            # read with the precision flag at the end of the guarded forward
        key = (ei.data_ptr(), E, n_nodes)
        if self.learn_edge_counts and not pad4 and not min4 and key not in self._edge_counts:
            self._edge_counts[key] = int(rowptr[-1].item())
        csr.edge_count = self._edge_counts.get(key, 0)
        return csr

    def csr_build_dual(self, edge_index: torch.Tensor, n_nodes: int, min4: Optional[bool] = None):
        """-> (plain CSR, 4-aligned CSR) of one square graph from one pass over the COO (morig_csr_build_dual).
        min4 (default: off; MORIG_EDGE_MIX=1 switches it on): the plain CSR's segments are filled up to 4 rows with copies of the self
        loop -- still a plain CSR for every kernel, and the form the H = 256 EdgeConv kernel takes WITHOUT 4-aligned segments (the
        mixed-quad kernel: correct and tested, measured no faster than the 4-aligned form -- DESIGN.md section 5.2 -- hence opt-in)."""
        if min4 is None:
            min4 = os.environ.get("MORIG_EDGE_MIX", "0") == "1"
        _need_gpu(edge_index)
        ei = edge_index if (edge_index.dtype == torch.int64 and edge_index.is_contiguous()) else edge_index.long().contiguous()
        E = ei.shape[1]
        dev = ei.device
        cap, cap4 = E + (4 if min4 else 1) * n_nodes, E + 4 * n_nodes
        rowptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
        rowptr4 = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
        src, dst = torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev)
        src4, dst4 = torch.empty(cap4, dtype=torch.int32, device=dev), torch.empty(cap4, dtype=torch.int32, device=dev)
        ws = torch.empty(2 * n_nodes + 1, dtype=torch.int32, device=dev)
        status = torch.empty(1, dtype=torch.int32, device=dev)
        check(self.lib.morig_csr_build_dual(_p(ei), E, n_nodes, _p(rowptr), _p(src), _p(dst), _p(rowptr4), _p(src4), _p(dst4), _p(ws),
                                            4 if min4 else 0, _p(status), _stream()), "morig_csr_build_dual")
        if getattr(self, "_csr_status", None) is not None:
            self._csr_status.append(status)
        key = (ei.data_ptr(), E, n_nodes)
        a = CSR(rowptr, src, dst, n_nodes, cap, status, min4=bool(min4))
        b = CSR(rowptr4, src4, dst4, n_nodes, cap4, status, quad=True)
        a.edge_count = b.edge_count = self._edge_counts.get(key, 0)
        return a, b

    def csr_from_slots(self, coo: torch.Tensor, n_nodes: int, max_nbrs: int, n_src: int) -> CSR:
        """the bipartite CSR of a ball-query slot table (``ball_query`` output): same result as
        ``csr_build(coo, n_nodes, n_src=n_src, skip_negative=True)``, built per target without atomics."""
        _need_gpu(coo)
        assert coo.dtype == torch.int64 and coo.is_contiguous() and coo.shape == (2, n_nodes * max_nbrs)
        dev = coo.device
        cap = n_nodes * (max_nbrs + 1)
        rowptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
        src = torch.empty(cap, dtype=torch.int32, device=dev)
        dst = torch.empty(cap, dtype=torch.int32, device=dev)
        cursor = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
        status = torch.empty(1, dtype=torch.int32, device=dev)
        check(self.lib.morig_csr_from_slots(_p(coo), n_nodes, max_nbrs, n_src, _p(rowptr), _p(src), _p(dst), _p(cursor),
                                            _p(status), _stream()), "morig_csr_from_slots")
        if getattr(self, "_csr_status", None) is not None:
            self._csr_status.append(status)
        return CSR(rowptr, src, dst, n_nodes, cap, status)

    # -- dense ----------------------------------------------------------------------------------
    def gemm(self, X: Mat, lin, relu: bool, Y: Optional[Mat] = None, rowbias: Optional[Mat] = None,
             seg: Optional[torch.Tensor] = None, pool: Optional[torch.Tensor] = None, affine: bool = True,
             x_split: bool = False, y_split: bool = False, x_tail: Optional[Mat] = None):
        """x_split / y_split: the X window is / the Y window shall be in the split-fp16 activation layout
        (include/morig_hip.h); only meaningful while ``self.fast``.
        x_tail: the last ``x_tail.cols`` columns of the input come from row (row % x_tail.rows) of this split-layout matrix instead of X
        (morig_gemm_args.X_tail: a replica-invariant block read from ONE copy); X then holds the first K - x_tail.cols columns.
        Only where ``gemm_takes_tail`` says so."""
        _need_gpu(X.base, lin.W)
        if x_split or y_split:
            if not self.fast:
                raise MorigNativeError("split-fp16 activations requested on the fp32 path (plan bug)")
            if lin.Wsplit is None:                    # weights outside the fp16 range: force the fp32 redo
                self._flag(X.base.device).fill_(1)
                return
        a = _args(GemmArgs)
        a.M, a.N, a.K = X.rows, lin.N, lin.K
        if x_tail is not None:
            assert x_split and X.cols + x_tail.cols == lin.K and x_tail.cols % 32 == 0, (X.cols, x_tail.cols, lin.K)
            a.X_tail, a.ld_tail, a.tail_rows, a.tail_cols = x_tail.ptr, x_tail.ld, x_tail.rows, x_tail.cols
        else:
            assert X.cols == lin.K, (X.cols, lin.K)
        a.X, a.ldx = X.ptr, X.ld
        a.W, a.ldw = lin.W.data_ptr(), lin.W.stride(0)
        a.bias = lin.bias.data_ptr() if lin.bias is not None else 0
        a.scale = lin.scale.data_ptr() if (affine and lin.scale is not None) else 0
        a.shift = lin.shift.data_ptr() if (affine and lin.shift is not None) else 0
        a.relu = 1 if relu else 0
        if rowbias is not None:
            a.rowbias, a.ld_rowbias = rowbias.ptr, rowbias.ld
        a.seg = seg.data_ptr() if seg is not None else 0
        if Y is not None:
            assert Y.rows == X.rows and Y.cols == lin.N
            a.Y, a.ldy = Y.ptr, Y.ld
        if pool is not None:
            assert pool.shape[1] >= lin.N and pool.is_contiguous()
            a.pool, a.ld_pool, a.n_seg = pool.data_ptr(), pool.stride(0), pool.shape[0]
        if getattr(lin, "Wsplit_bf16", None) is not None:
            # the bf16 split (no range guard: W_split without an overflow word, include/morig_hip.h): whatever self.precision says
            assert pool is None and not x_split and not y_split
            a.W_split, a.overflow, a.w_split_format = lin.Wsplit_bf16.data_ptr(), 0, 1     # MORIG_SPLIT_BF16
        elif self.fast and lin.Wsplit is not None:
            a.W_split, a.overflow = lin.Wsplit.data_ptr(), self._flag(X.base.device).data_ptr()
        elif self.exact_arith == "bf16x6" and not x_split and not y_split:
            a.w_split_format = 2                                                             # MORIG_SPLIT_BF16X6, no image: split in the kernel
        a.x_split, a.y_split = int(x_split), int(y_split)
        check(self.lib.morig_gemm(C.byref(a), _stream()), "morig_gemm")

    def gemm_takes_tail(self, lin, Y: Mat, n_tail_cols: int) -> bool:
        """May ``gemm(..., x_split=True, x_tail=...)`` be used for this layer and output window? Mirrors the library's kernel choice
        (csrc/gemm_dma.hip: K tails exist on the 256 x 256 LDS-DMA store kernel only); MORIG_GEMM_TAIL=0 switches the tails off (A/B runs:
        the block is then copied into every replica's row, the round-5 plan)."""
        return (self.fast and os.environ.get("MORIG_GEMM_TAIL", "1") != "0" and os.environ.get("MORIG_NO_DMA") is None and
                lin.Wsplit is not None and lin.N % 256 == 0 and lin.K % 32 == 0 and n_tail_cols % 32 == 0 and lin.K > n_tail_cols and
                Y.ptr % 16 == 0 and Y.ld % 4 == 0 and os.environ.get("MORIG_GEMM_TR", "1") != "0" and
                os.environ.get("MORIG_DMA_TILE", "256") == "256")

    def pack_tails(self, src: torch.Tensor, col_a, col_b, wa: int, wb: int) -> torch.Tensor:
        """-> [n_tails, rows, 32] split-fp16 chunks: tail t, row v = [src[v, col_a[t]:+wa] | src[v, col_b[t]:+wb] | 0] (morig_pack_tails),
        ONE launch for every tail of a forward."""
        _need_gpu(src)
        assert src.dim() == 2 and src.dtype == torch.float32 and src.stride(1) == 1 and len(col_a) == len(col_b) <= 8
        n, rows = len(col_a), src.shape[0]
        out = torch.empty((n, rows, 32), dtype=torch.float32, device=src.device)
        ca, cb = (C.c_int32 * n)(*col_a), (C.c_int32 * n)(*col_b)
        check(self.lib.morig_pack_tails(_p(src), src.stride(0), rows, ca, cb, wa, wb, n, _p(out), rows * 32,
                                        self._flag(src.device).data_ptr(), _stream()), "morig_pack_tails")
        return out

    # -- fused edge conv ----------------------------------------------------------------------------
    def edgeconv(self, A: Mat, B: Mat, csr: CSR, ec, out: Mat, replicas: int = 1,
                 in_rep_stride: int = 0, out_rep_stride: int = 0, out_split: bool = False):
        """out_split: `out` (a chunk-aligned window, H % 32 == 0) is written in the split-fp16 activation layout; only where
        ``edgeconv_can_split_out`` says so for the same arguments (MORIG_E_UNSUPPORTED otherwise)."""
        _need_gpu(A.base, B.base, out.base)
        a = self._edge_args(A, B, csr, ec, out, replicas, in_rep_stride, out_rep_stride)
        a.out_split = 1 if out_split else 0
        check(self.lib.morig_edgeconv(C.byref(a), _stream()), "morig_edgeconv")

    def edgeconv_pair(self, first: dict, second: dict):
        """Two EdgeConv launches that write disjoint column blocks of the same rows and do not read each other's results (the template-
        and the geodesic-graph EdgeConv of a unit): ``first`` / ``second`` are the keyword arguments of ``edgeconv``. Same results as the
        two calls; their boundary passes share launches (morig_edgeconv_args.init_with / split_with: one identity pass in front of both
        kernels, one conversion pass behind both when both store split rows) -- 2 x 2 small launches less per unit.
        MORIG_EDGE_PAIR=0: two plain calls (A/B runs)."""
        if not PAIR_PASSES:
            self.edgeconv(**first)
            self.edgeconv(**second)
            return
        calls = []
        for kw in (first, second):
            kw = dict(kw)
            sp = bool(kw.pop("out_split", False))
            _need_gpu(kw["A"].base, kw["B"].base, kw["out"].base)
            a = self._edge_args(kw["A"], kw["B"], kw["csr"], kw["ec"], kw["out"], kw.get("replicas", 1), kw.get("in_rep_stride", 0),
                                kw.get("out_rep_stride", 0))
            a.out_split = 1 if sp else 0
            calls.append(a)
        a1, a2 = calls
        a1.init_with, a2.skip_init = C.addressof(a2), 1
        if a1.out_split and a2.out_split:
            a1.skip_split, a2.split_with = 1, C.addressof(a1)
        check(self.lib.morig_edgeconv(C.byref(a1), _stream()), "morig_edgeconv")
        check(self.lib.morig_edgeconv(C.byref(a2), _stream()), "morig_edgeconv")

    def edgeconv_x3_pair(self, first: dict, second: dict):
        """the same for two ``edgeconv_x3`` launches (keyword arguments of ``edgeconv_x3``): one identity pass for both"""
        if not PAIR_PASSES:
            self.edgeconv_x3(**first)
            self.edgeconv_x3(**second)
            return
        a1, a2 = (self._x3_args(**kw) for kw in (first, second))
        a1.init_with, a2.skip_init = C.addressof(a2), 1
        check(self.lib.morig_edgeconv_x3(C.byref(a1), _stream()), "morig_edgeconv_x3")
        check(self.lib.morig_edgeconv_x3(C.byref(a2), _stream()), "morig_edgeconv_x3")

    def edgeconv_can_split_out(self, A: Mat, B: Mat, csr: CSR, ec, out: Mat, replicas: int = 1,
                               in_rep_stride: int = 0, out_rep_stride: int = 0) -> bool:
        """Would ``edgeconv(..., out_split=True)`` run for these arguments? (kernel choice, alignment, environment switches: the library
        decides, morig_edgeconv_can_split_out; nothing is launched)"""
        a = self._edge_args(A, B, csr, ec, out, replicas, in_rep_stride, out_rep_stride)
        return bool(self.lib.morig_edgeconv_can_split_out(C.byref(a)))

    def _edge_args(self, A, B, csr, ec, out, replicas, in_rep_stride, out_rep_stride):
        a = _args(EdgeConvArgs)
        a.H = ec.H
        a.n_nodes, a.replicas = csr.n_nodes, replicas
        a.in_rep_stride, a.out_rep_stride = in_rep_stride, out_rep_stride
        a.A, a.lda = A.ptr, A.ld
        a.B, a.ldb = B.ptr, B.ld
        a.rowptr, a.src_sorted, a.dst_sorted = csr.rowptr.data_ptr(), csr.src.data_ptr(), csr.dst.data_ptr()
        a.edge_capacity, a.edge_count = csr.capacity, csr.edge_count
        if ec.s1 is not None:
            a.s1, a.t1 = ec.s1.data_ptr(), ec.t1.data_ptr()
        a.W2, a.ldw = ec.W2.data_ptr(), ec.W2.stride(0)
        a.b2, a.s2, a.t2 = ec.b2.data_ptr(), ec.s2.data_ptr(), ec.t2.data_ptr()
        a.out, a.ldo = out.ptr, out.ld
        a.quad_aligned = 1 if csr.quad else 0
        a.seg_min4 = 1 if getattr(csr, "min4", False) else 0
        a.exact_arith = 1 if self.exact_arith == "bf16x6" else 0
        if self.fast and ec.W2split is not None:
            a.W2_split, a.overflow = ec.W2split.data_ptr(), self._flag(A.base.device).data_ptr()
        return a

    def edgeconv_x3(self, X: Mat, first, csr: CSR, ec, out: Mat, replicas: int = 1, in_rep_stride: int = 0, out_rep_stride: int = 0):
        """EdgeConv on a 3-channel vertex input with the first Linear evaluated in the kernel (morig_edgeconv_x3). X: [rows, >= 4]
        window starting at column 0 of 16-byte aligned rows; first = (W1a [32, 4], W1b [32, 4], b1 [32]) from packing.pack_first_x3."""
        a = self._x3_args(X, first, csr, ec, out, replicas, in_rep_stride, out_rep_stride)
        check(self.lib.morig_edgeconv_x3(C.byref(a), _stream()), "morig_edgeconv_x3")

    def _x3_args(self, X: Mat, first, csr: CSR, ec, out: Mat, replicas: int = 1, in_rep_stride: int = 0, out_rep_stride: int = 0):
        _need_gpu(X.base, out.base)
        assert ec.H == 32 and ec.s1 is None and X.col0 % 4 == 0 and X.ld % 4 == 0
        a = _args(EdgeConvX3Args)
        a.H = 32
        a.n_nodes, a.replicas = csr.n_nodes, replicas
        a.in_rep_stride, a.out_rep_stride = in_rep_stride, out_rep_stride
        a.X, a.ldx = X.ptr, X.ld
        a.W1a, a.W1b, a.b1 = first[0].data_ptr(), first[1].data_ptr(), first[2].data_ptr()
        a.rowptr, a.src_sorted, a.dst_sorted = csr.rowptr.data_ptr(), csr.src.data_ptr(), csr.dst.data_ptr()
        a.edge_capacity, a.edge_count = csr.capacity, csr.edge_count
        a.W2, a.ldw = ec.W2.data_ptr(), ec.W2.stride(0)
        a.b2, a.s2, a.t2 = ec.b2.data_ptr(), ec.s2.data_ptr(), ec.t2.data_ptr()
        a.out, a.ldo = out.ptr, out.ld
        if self.fast and ec.W2split is not None:
            a.W2_split, a.overflow = ec.W2split.data_ptr(), self._flag(X.base.device).data_ptr()
        return a

    def edge_hidden(self, A: Mat, B: Mat, csr: CSR, ec, Z: Mat):
        """per-edge hidden activations Z [capacity, H] (rows >= E' untouched)."""
        _need_gpu(A.base, B.base, Z.base)
        assert Z.rows == csr.capacity and Z.cols == ec.H
        a = self._edge_args(A, B, csr, ec, Z, 1, 0, 0)
        check(self.lib.morig_edge_hidden(C.byref(a), _stream()), "morig_edge_hidden")

    def segmax_gemm(self, X: Mat, lin, relu: bool, csr: CSR, out: Mat):
        _need_gpu(X.base, out.base)
        assert X.rows == csr.capacity and X.cols == lin.K and out.rows == csr.n_nodes and out.cols == lin.N
        a = _args(SegmaxArgs)
        a.N, a.K = lin.N, lin.K
        a.X, a.ldx = X.ptr, X.ld
        a.W, a.ldw = lin.W.data_ptr(), lin.W.stride(0)
        a.bias = lin.bias.data_ptr() if lin.bias is not None else 0
        a.scale = lin.scale.data_ptr() if lin.scale is not None else 0
        a.shift = lin.shift.data_ptr() if lin.shift is not None else 0
        a.relu = 1 if relu else 0
        a.rowptr, a.dst_sorted, a.n_nodes = csr.rowptr.data_ptr(), csr.dst.data_ptr(), csr.n_nodes
        a.edge_capacity, a.edge_count = csr.capacity, csr.edge_count
        a.out, a.ldo = out.ptr, out.ld
        if self.fast and lin.Wsplit is not None:
            a.W_split, a.overflow = lin.Wsplit.data_ptr(), self._flag(X.base.device).data_ptr()
        check(self.lib.morig_segmax_gemm(C.byref(a), _stream()), "morig_segmax_gemm")

    def pointconv_can_fuse(self, pk, max_nbrs: int) -> bool:
        """the one-launch PointConv (csrc/pointconv_fused.hip) covers this packed local_nn? (split-fp16 arithmetic only)"""
        l3 = pk.get("fused")
        return bool(self.fast and l3 is not None and l3.Wsplit is not None and pk["edge"].W2split is not None and
                    max_nbrs == 64 and pk["edge"].s1 is None and os.environ.get("MORIG_POINTCONV_FUSED", "1") != "0")

    def pointconv_fused(self, A: Mat, B: Mat, coo: torch.Tensor, max_nbrs: int, pk, out: Mat):
        """out[c] = max over the slot table's kept edges + the self loop (c, c) of the 3-layer PointConv message
        (== csr_from_slots -> edge_hidden -> segmax_gemm); coo: the ball_query slot table [2, n_centres * max_nbrs]."""
        _need_gpu(A.base, B.base, out.base, coo)
        ec, l3 = pk["edge"], pk["fused"]
        assert coo.dtype == torch.int64 and coo.is_contiguous() and coo.shape == (2, A.rows * max_nbrs)
        assert A.cols == B.cols == ec.H and out.rows == A.rows and out.cols == l3.N and B.rows >= A.rows
        dev = A.base.device
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        a = _args(PointConvArgs)
        a.A, a.lda, a.B, a.ldb = A.ptr, A.ld, B.ptr, B.ld
        a.slots, a.max_nbrs, a.n_centres, a.n_src = coo.data_ptr(), max_nbrs, A.rows, B.rows
        a.H, a.H3 = ec.H, l3.N
        a.W2_split, a.ldw2, a.b2 = ec.W2split.data_ptr(), ec.W2split.stride(0), ec.b2.data_ptr()
        a.W3_split, a.ldw3, a.b3 = l3.Wsplit.data_ptr(), l3.Wsplit.stride(0), l3.bias.data_ptr()
        a.s3 = l3.scale.data_ptr() if l3.scale is not None else 0
        a.t3 = l3.shift.data_ptr() if l3.shift is not None else 0
        a.relu3 = 1
        a.out, a.ldo = out.ptr, out.ld
        a.overflow, a.status = self._flag(dev).data_ptr(), status.data_ptr()
        check(self.lib.morig_pointconv_fused(C.byref(a), _stream()), "morig_pointconv_fused")
        if getattr(self, "_csr_status", None) is not None:
            self._csr_status.append(status)            # a slot naming a source >= n_src: raised at the end of the guarded forward
        elif int(status.item()) != 0:
            raise MorigNativeError("morig_pointconv_fused: the slot table names a source row outside B")

    # -- point clouds ---------------------------------------------------------------------------------
    def fps(self, pos: Mat, ptr: torch.Tensor, out_ptr: torch.Tensor, start: Optional[torch.Tensor], n_clouds: int,
            max_cloud_points: int, n_samples: int) -> torch.Tensor:
        _need_gpu(pos.base, ptr, out_ptr)
        idx = torch.empty(n_samples, dtype=torch.int32, device=pos.base.device)
        check(self.lib.morig_fps(pos.ptr, pos.ld, _p(ptr), _p(out_ptr), _p(start), n_clouds, max_cloud_points, _p(idx), _stream()),
              "morig_fps")
        return idx

    def ball_query(self, x: Mat, ptr_x: torch.Tensor, y: Mat, ptr_y: torch.Tensor, n_clouds: int, radius: float,
                   max_nbrs: int) -> torch.Tensor:
        _need_gpu(x.base, y.base, ptr_x, ptr_y)
        coo = torch.empty((2, y.rows * max_nbrs), dtype=torch.int64, device=x.base.device)
        check(self.lib.morig_ball_query(x.ptr, x.ld, _p(ptr_x), y.ptr, y.ld, _p(ptr_y), n_clouds, y.rows, float(radius),
                                        max_nbrs, _p(coo), _stream()), "morig_ball_query")
        return coo

    # -- train-mode forward support (csrc/train_ops.hip) ---------------------------------
    def col_stats(self, X: Mat, rows_dev: Optional[torch.Tensor] = None):
        """-> (mean [cols], biased var [cols], count [1]) of the rows of X (rows_dev: int32 [1] on the device = live row count)."""
        _need_gpu(X.base)
        dev = X.base.device
        slabs = (max(X.rows, 1) + 255) // 256
        ws = torch.empty(slabs * 2 * X.cols, dtype=torch.float64, device=dev)
        mean = torch.empty(X.cols, dtype=torch.float32, device=dev)
        var = torch.empty(X.cols, dtype=torch.float32, device=dev)
        cnt = torch.empty(1, dtype=torch.float32, device=dev)
        check(self.lib.morig_col_stats(X.ptr, X.ld, X.rows, _p(rows_dev), X.cols, C.c_void_p(ws.data_ptr()), ws.numel(), _p(mean),
                                       _p(var), _p(cnt), _stream()), "morig_col_stats")
        return mean, var, cnt

    def col_affine(self, X: Mat, scale: torch.Tensor, shift: torch.Tensor, rows_dev: Optional[torch.Tensor] = None,
                   out: Optional[Mat] = None):
        """out = scale * X + shift per column (in place without ``out``)"""
        _need_gpu(X.base, scale, shift)
        assert scale.numel() >= X.cols and shift.numel() >= X.cols and scale.dtype == shift.dtype == torch.float32
        assert out is None or (out.rows == X.rows and out.cols == X.cols)
        check(self.lib.morig_col_affine(X.ptr, X.ld, X.rows, _p(rows_dev), X.cols, _p(scale), _p(shift), out.ptr if out is not None else 0,
                                        out.ld if out is not None else 0, _stream()), "morig_col_affine")

    def bn_finalize(self, bn, mean: torch.Tensor, var: torch.Tensor, count: torch.Tensor):
        """BatchNorm1d ``bn`` in training mode after its column statistics, ONE launch: -> (s, t, rstd); the running buffers and
        num_batches_tracked move in place (``bn.momentum`` must be a number)."""
        _need_gpu(mean, var, count)
        n = mean.numel()
        dev = mean.device
        out = torch.empty((3, n), dtype=torch.float32, device=dev)
        track = bn.track_running_stats and bn.running_mean is not None
        g = bn.weight.detach() if bn.weight is not None else None
        b = bn.bias.detach() if bn.bias is not None else None
        for x in (g, b, bn.running_mean if track else None, bn.running_var if track else None):
            assert x is None or (x.dtype == torch.float32 and x.is_contiguous() and x.device == dev)
        assert mean.dtype == var.dtype == count.dtype == torch.float32 and mean.is_contiguous() and var.is_contiguous()
        check(self.lib.morig_bn_finalize(_p(mean), _p(var), _p(count), _p(g), _p(b), float(bn.eps), float(bn.momentum if track else 0.0),
                                         _p(bn.running_mean) if track else None, _p(bn.running_var) if track else None,
                                         _p(bn.num_batches_tracked) if track else None, _p(out[0]), _p(out[1]), _p(out[2]), n, _stream()),
              "morig_bn_finalize")
        return out[0], out[1], out[2]

    def edge_gather_relu(self, A: Mat, B: Mat, csr: CSR, Z: Mat, want_stats: bool = False):
        """want_stats: -> (mean, biased var, count [1]) of the live rows of Z, as ``col_stats(Z, rows_dev=E')``, from the same pass"""
        _need_gpu(A.base, B.base, Z.base)
        assert Z.rows == csr.capacity and A.cols == B.cols == Z.cols
        ws = st = None
        if want_stats:
            dev = Z.base.device
            ws = torch.empty(((csr.capacity + 255) // 256) * 2 * Z.cols, dtype=torch.float64, device=dev)
            st = torch.empty(2 * Z.cols + 1, dtype=torch.float32, device=dev)
        check(self.lib.morig_edge_gather_relu(A.ptr, A.ld, B.ptr, B.ld, _p(csr.rowptr), csr.n_nodes, _p(csr.src), _p(csr.dst),
                                              csr.capacity, Z.cols, Z.ptr, Z.ld, _p(ws), ws.numel() if want_stats else 0,
                                              _p(st), _p(st[Z.cols:]) if want_stats else _p(None),
                                              _p(st[2 * Z.cols:]) if want_stats else _p(None), _stream()), "morig_edge_gather_relu")
        return (st[:Z.cols], st[Z.cols:2 * Z.cols], st[2 * Z.cols:]) if want_stats else None

    def segmax_affine(self, Z: Mat, rowptr: torch.Tensor, n_segments: int, out: Mat, scale: Optional[torch.Tensor] = None,
                      shift: Optional[torch.Tensor] = None):
        _need_gpu(Z.base, rowptr, out.base)
        assert rowptr.dtype == torch.int32 and out.rows == n_segments and out.cols == Z.cols
        check(self.lib.morig_segmax_affine(Z.ptr, Z.ld, _p(rowptr), n_segments, Z.cols, _p(scale), _p(shift), out.ptr, out.ld,
                                           _stream()), "morig_segmax_affine")

    # -- train-mode backward support (csrc/train_bwd.hip; SURVEY 8 f-4, backward half) --------------------------------
    def bn_backward_stats(self, dz: Mat, y: Optional[Mat] = None, mean: Optional[torch.Tensor] = None,
                          rstd: Optional[torch.Tensor] = None, rows_dev: Optional[torch.Tensor] = None):
        """-> (sum_dz [cols], sum_dz_xhat [cols] or None): dbeta / dgamma of a training-mode BatchNorm1d with input y
        (y None: the plain column sum = dbias)."""
        _need_gpu(dz.base)
        dev = dz.base.device
        slabs = (max(dz.rows, 1) + 255) // 256
        ws = torch.empty(slabs * 2 * dz.cols, dtype=torch.float64, device=dev)
        sdz = torch.empty(dz.cols, dtype=torch.float32, device=dev)
        sdzx = torch.empty(dz.cols, dtype=torch.float32, device=dev) if y is not None else None
        check(self.lib.morig_bn_backward_stats(dz.ptr, dz.ld, y.ptr if y is not None else 0, y.ld if y is not None else 0, dz.rows,
                                               _p(rows_dev), dz.cols, _p(mean), _p(rstd), C.c_void_p(ws.data_ptr()), ws.numel(),
                                               _p(sdz), _p(sdzx), _stream()), "morig_bn_backward_stats")
        return sdz, sdzx

    def bn_relu_backward(self, dz: Mat, y: Mat, mean, rstd, gamma, sum_dz, sum_dzx, du: Mat, rows_dev: Optional[torch.Tensor] = None,
                         want_sum: bool = False):
        """want_sum: -> the column sums of du (float32 [cols], = the bias gradient of the Linear in front), from the same pass"""
        _need_gpu(dz.base, y.base, du.base)
        assert dz.rows == y.rows == du.rows and dz.cols == y.cols == du.cols
        ws = sdu = None
        if want_sum:
            dev = du.base.device
            ws = torch.empty(((max(dz.rows, 1) + 255) // 256) * 2 * dz.cols, dtype=torch.float64, device=dev)
            sdu = torch.empty(dz.cols, dtype=torch.float32, device=dev)
        check(self.lib.morig_bn_relu_backward(dz.ptr, dz.ld, y.ptr, y.ld, dz.rows, _p(rows_dev), dz.cols, _p(mean), _p(rstd), _p(gamma),
                                              _p(sum_dz), _p(sum_dzx), du.ptr, du.ld, _p(ws), ws.numel() if want_sum else 0, _p(sdu),
                                              _stream()), "morig_bn_relu_backward")
        return sdu

    def segmax_affine_arg(self, Z: Mat, rowptr: torch.Tensor, n_segments: int, out: Mat, scale=None, shift=None, want_zwin: bool = False):
        """segmax_affine + the winning row per (segment, column): int32 [n_segments, cols], -1 for empty segments. want_zwin: also the
        winners' values in front of the affine, float32 [n_segments, cols] -> (arg, zwin)."""
        _need_gpu(Z.base, rowptr, out.base)
        assert rowptr.dtype == torch.int32 and out.rows == n_segments and out.cols == Z.cols
        arg = torch.empty((n_segments, Z.cols), dtype=torch.int32, device=Z.base.device)
        zwin = torch.empty((n_segments, Z.cols), dtype=torch.float32, device=Z.base.device) if want_zwin else None
        check(self.lib.morig_segmax_affine_arg(Z.ptr, Z.ld, _p(rowptr), n_segments, Z.cols, _p(scale), _p(shift), out.ptr, out.ld,
                                               _p(arg), arg.stride(0), _p(zwin), zwin.stride(0) if want_zwin else 0, _stream()),
              "morig_segmax_affine_arg")
        return (arg, zwin) if want_zwin else arg

    def segmax_bn_backward_stats(self, dout: Mat, arg: torch.Tensor, Z: Optional[Mat], mean, rstd, zwin: Optional[torch.Tensor] = None):
        """zwin ([n_seg, cols], from segmax_affine_arg): the coalesced kernel; without it the winners' rows are gathered from Z."""
        _need_gpu(dout.base, arg, Z.base if Z is not None else zwin)
        dev = dout.base.device
        n_seg, cols = dout.rows, dout.cols
        assert arg.shape == (n_seg, cols) and (Z is None or Z.cols == cols) and (zwin is None or zwin.shape == (n_seg, cols))
        slabs = (n_seg + 255) // 256
        ws = torch.empty(slabs * 2 * cols, dtype=torch.float64, device=dev)
        sdz = torch.empty(cols, dtype=torch.float32, device=dev)
        sdzx = torch.empty(cols, dtype=torch.float32, device=dev)
        check(self.lib.morig_segmax_bn_backward_stats(dout.ptr, dout.ld, _p(arg), arg.stride(0), Z.ptr if Z is not None else 0,
                                                      Z.ld if Z is not None else 0, _p(zwin), zwin.stride(0) if zwin is not None else 0,
                                                      n_seg, cols, _p(mean), _p(rstd), C.c_void_p(ws.data_ptr()), ws.numel(), _p(sdz),
                                                      _p(sdzx), _stream()), "morig_segmax_bn_backward_stats")
        return sdz, sdzx

    def segmax_bn_relu_backward(self, dout: Mat, arg: torch.Tensor, Z: Mat, rowptr: torch.Tensor, seg_of_row: torch.Tensor, mean, rstd,
                                gamma, sum_dz, sum_dzx, du: Mat, relu: bool = True, want_sum: bool = False):
        """want_sum: -> the column sums of du over the live rows (float32 [cols]), from the same pass."""
        _need_gpu(dout.base, arg, Z.base, du.base, rowptr, seg_of_row)
        assert Z.rows == du.rows and Z.cols == du.cols == dout.cols and seg_of_row.dtype == torch.int32
        ws = sdu = None
        if want_sum:
            dev = du.base.device
            ws = torch.empty(((Z.rows + 255) // 256) * 2 * Z.cols, dtype=torch.float64, device=dev)
            sdu = torch.empty(Z.cols, dtype=torch.float32, device=dev)
        check(self.lib.morig_segmax_bn_relu_backward(dout.ptr, dout.ld, _p(arg), arg.stride(0), Z.ptr, Z.ld, _p(rowptr), dout.rows,
                                                     _p(seg_of_row), Z.rows, Z.cols, _p(mean), _p(rstd), _p(gamma), _p(sum_dz), _p(sum_dzx),
                                                     1 if relu else 0, du.ptr, du.ld, _p(ws), ws.numel() if want_sum else 0, _p(sdu),
                                                     _stream()), "morig_segmax_bn_relu_backward")
        return sdu

    def edge_bn_scatter_backward(self, dG: Mat, Y: Optional[Mat], csr: CSR, n_src: int, dA: Mat, dB: Mat, mean=None, rstd=None, gamma=None,
                                 sum_dz=None, sum_dzx=None, ZA: Optional[Mat] = None, ZB: Optional[Mat] = None):
        """bn_relu_backward + edge_scatter_backward in one pass pair, deterministic (no atomics, the per-edge gradient is not stored):
        dA[v] / dB[u] = sums of d[e] = [Y > 0] gamma rstd (dG - sum_dz / n - xhat sum_dzx / n) into / out of a vertex; mean=None: d = dG."""
        _need_gpu(dG.base, dA.base, dB.base)
        assert dA.rows == csr.n_nodes and dB.rows == n_src and dA.cols == dB.cols == dG.cols
        assert mean is None or ZA is not None or (Y is not None and Y.cols == dG.cols and Y.rows == dG.rows)
        assert (ZA is None) == (ZB is None) and (ZA is None or (ZA.rows == csr.n_nodes and ZB.rows == n_src and ZA.cols == ZB.cols == dG.cols))
        rowptr_t, perm_t = csr.transposed(n_src)
        check(self.lib.morig_edge_bn_scatter_backward(dG.ptr, dG.ld, Y.ptr if Y is not None else 0, Y.ld if Y is not None else 0,
                                                      _p(csr.rowptr), _p(rowptr_t), _p(perm_t), csr.n_nodes, n_src, dG.cols, _p(mean),
                                                      _p(rstd), _p(gamma), _p(sum_dz), _p(sum_dzx), dA.ptr, dA.ld, dB.ptr, dB.ld,
                                                      ZA.ptr if ZA is not None else 0, ZA.ld if ZA is not None else 0,
                                                      ZB.ptr if ZB is not None else 0, ZB.ld if ZB is not None else 0,
                                                      _p(csr.src) if ZA is not None else None, _p(csr.dst) if ZA is not None else None,
                                                      _stream()), "morig_edge_bn_scatter_backward")

    def edge_scatter_backward(self, dG: Mat, csr: CSR, n_src: int, dA: Mat, dB: Mat):
        """backward of Z[e] = A[dst_e] + B[src_e] over the live edges of ``csr``."""
        _need_gpu(dG.base, dA.base, dB.base)
        assert dA.rows == csr.n_nodes and dB.rows == n_src and dA.cols == dB.cols == dG.cols
        check(self.lib.morig_edge_scatter_backward(dG.ptr, dG.ld, _p(csr.rowptr), _p(csr.src), csr.n_nodes, n_src, dG.cols, dA.ptr, dA.ld,
                                                   dB.ptr, dB.ld, _stream()), "morig_edge_scatter_backward")

    def edge_bn_sums_from_products(self, M: torch.Tensor, db2: torch.Tensor, W2: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor):
        """-> (sum_dz, sum_dzx) of the first edge layer's BatchNorm from M = dU2^T Z1, db2 and W2 (no pass over the edges)"""
        _need_gpu(M, db2, W2, mean, rstd)
        h_out, h_in = W2.shape
        assert M.shape == (h_out, h_in) and M.stride(1) == 1 and W2.stride(1) == 1 and db2.numel() >= h_out and db2.is_contiguous()
        assert all(t.dtype == torch.float32 for t in (M, db2, W2, mean, rstd)) and mean.numel() >= h_in and rstd.numel() >= h_in
        out = torch.empty((2, h_in), dtype=torch.float32, device=M.device)
        check(self.lib.morig_edge_bn_sums_from_products(_p(M), M.stride(0), _p(db2), _p(W2), W2.stride(0), _p(mean), _p(rstd), h_out, h_in,
                                                        _p(out[0]), _p(out[1]), _stream()), "morig_edge_bn_sums_from_products")
        return out[0], out[1]

    def gemm_tn(self, A: Mat, B: Mat, out: Optional[Mat] = None, rows_dev: Optional[torch.Tensor] = None,
                b_shift: Optional[torch.Tensor] = None) -> torch.Tensor:
        """A^T B over the rows: [A.cols, B.cols] (the weight gradient dU^T X). b_shift [B.cols]: every row of B is centred on it first,
        A^T (B - 1 b_shift^T)."""
        _need_gpu(A.base, B.base)
        if b_shift is not None:
            assert b_shift.dtype == torch.float32 and b_shift.is_contiguous() and b_shift.numel() >= B.cols
        assert A.rows == B.rows
        dev = A.base.device
        if out is None:
            out = Mat.of(torch.empty((A.cols, B.cols), dtype=torch.float32, device=dev))
        n_ws = int(self.lib.morig_gemm_tn_workspace(A.rows, A.cols, B.cols))
        ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dev)
        check(self.lib.morig_gemm_tn_shift(A.ptr, A.ld, B.ptr, B.ld, _p(b_shift), A.rows, _p(rows_dev), A.cols, B.cols, _p(ws), ws.numel(),
                                           out.ptr, out.ld, _stream()), "morig_gemm_tn_shift")
        return out.base

    def radius_sample(self, x: Mat, y: Mat, radius: float, max_nbrs: int, seed: int):
        """radius_cpu's neighbour table: (slot table int64 [2, ny * max_nbrs] (-1 = unused), hits per row int32 [ny])."""
        _need_gpu(x.base, y.base)
        dev = x.base.device
        coo = torch.empty((2, y.rows * max_nbrs), dtype=torch.int64, device=dev)
        counts = torch.empty(y.rows, dtype=torch.int32, device=dev)
        check(self.lib.morig_radius_sample(x.ptr, x.ld, x.rows, y.ptr, y.ld, y.rows, float(radius), max_nbrs, int(seed) & 0xFFFFFFFF,
                                           _p(coo), _p(counts), _stream()), "morig_radius_sample")
        return coo, counts

    def geo_ball_graph(self, pos: Optional[Mat], mesh_ptr: Optional[torch.Tensor], radius: float, max_nn: int, seed: int,
                       self_loops: bool = False, dist: Optional[torch.Tensor] = None):
        """get_geo_edges (data_proc/common_ops.py:214-226) on the device -> (edge_index int64 [2, E] rows [i, member],
        members int32 [n] uncapped ball sizes). Positions variant: pos [n, >=3] + mesh_ptr int32 [B + 1]; distance variant:
        dist float64 [n, n] of one mesh. ONE host read (the edge count sizes the result)."""
        if dist is not None:
            _need_gpu(dist)
            assert dist.dtype == torch.float64 and dist.dim() == 2 and dist.shape[0] == dist.shape[1] and dist.stride(1) == 1
            n, dev = dist.shape[0], dist.device
        else:
            _need_gpu(pos.base, mesh_ptr)
            assert mesh_ptr.dtype == torch.int32
            n, dev = pos.rows, pos.base.device
        slots = torch.empty((n, max_nn), dtype=torch.int32, device=dev)
        counts = torch.empty(n, dtype=torch.int32, device=dev)
        members = torch.empty(n, dtype=torch.int32, device=dev)
        offsets = torch.empty(n + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(max(1, (n + 2047) // 2048), dtype=torch.int32, device=dev)
        sd = int(seed) & 0xFFFFFFFF
        if dist is not None:
            check(self.lib.morig_geo_ball_graph_dist(_p(dist), dist.stride(0), n, float(radius), max_nn, sd, _p(slots), _p(counts),
                                                     _p(members), _p(offsets), _p(ws), _stream()), "morig_geo_ball_graph_dist")
        else:
            check(self.lib.morig_geo_ball_graph(pos.ptr, pos.ld, _p(mesh_ptr), mesh_ptr.numel() - 1, n, float(radius), max_nn, sd,
                                                _p(slots), _p(counts), _p(members), _p(offsets), _p(ws), _stream()),
                  "morig_geo_ball_graph")
        n_out = int(offsets[n].item()) + (n if self_loops else 0)
        coo = torch.empty((2, n_out), dtype=torch.int64, device=dev)
        if n_out:
            check(self.lib.morig_geo_ball_fill(_p(slots), _p(offsets), n, max_nn, 1 if self_loops else 0, _p(coo), n_out, _stream()),
                  "morig_geo_ball_fill")
        return coo, members

    def knn_interpolate(self, feat: Mat, pos_x: Mat, ptr_x: torch.Tensor, pos_y: Mat, ptr_y: torch.Tensor, n_clouds: int,
                        max_targets_per_cloud: int, k: int, out: Mat):
        _need_gpu(feat.base, pos_x.base, pos_y.base, out.base)
        nt = pos_y.rows
        dev = feat.base.device
        idx = torch.empty((nt, 3), dtype=torch.int32, device=dev)
        wgt = torch.empty((nt, 3), dtype=torch.float32, device=dev)
        assert out.rows == nt and out.cols == feat.cols
        check(self.lib.morig_knn_interpolate(feat.ptr, feat.ld, feat.cols, pos_x.ptr, pos_x.ld, _p(ptr_x), pos_y.ptr, pos_y.ld,
                                             _p(ptr_y), n_clouds, nt, max_targets_per_cloud, k, _p(idx), _p(wgt), out.ptr, out.ld,
                                             _stream()), "morig_knn_interpolate")

    def knn_search(self, pos_x: Mat, ptr_x: torch.Tensor, pos_y: Mat, ptr_y: torch.Tensor, n_clouds: int,
                   max_targets_per_cloud: int, k: int):
        """the geometry half of knn_interpolate: -> (idx [ny, 3] int32, wgt [ny, 3]) for ``knn_apply``."""
        _need_gpu(pos_x.base, pos_y.base)
        nt = pos_y.rows
        dev = pos_x.base.device
        idx = torch.empty((nt, 3), dtype=torch.int32, device=dev)
        wgt = torch.empty((nt, 3), dtype=torch.float32, device=dev)
        check(self.lib.morig_knn_search(pos_x.ptr, pos_x.ld, _p(ptr_x), pos_y.ptr, pos_y.ld, _p(ptr_y), n_clouds, nt,
                                        max_targets_per_cloud, k, _p(idx), _p(wgt), _stream()), "morig_knn_search")
        return idx, wgt

    def knn_apply(self, feat: Mat, nn, out: Mat):
        """out[t] = sum_s w_s feat[idx_s] / sum_s w_s with (idx, wgt) = nn from ``knn_search``."""
        idx, wgt = nn
        _need_gpu(feat.base, out.base, idx, wgt)
        assert out.rows == idx.shape[0] and out.cols == feat.cols
        check(self.lib.morig_knn_apply(feat.ptr, feat.ld, feat.cols, _p(idx), _p(wgt), out.rows, out.ptr, out.ld, _stream()),
              "morig_knn_apply")

    def cosine_nn(self, v: Mat, ptr_v: torch.Tensor, p: Mat, ptr_p: torch.Tensor, n_clouds: int, max_rows_per_cloud: int):
        _need_gpu(v.base, p.base)
        dev = v.base.device
        nn = torch.empty(v.rows, dtype=torch.int32, device=dev)
        sim = torch.empty(v.rows, dtype=torch.float32, device=dev)
        check(self.lib.morig_cosine_nn(v.ptr, v.ld, _p(ptr_v), p.ptr, p.ld, _p(ptr_p), n_clouds, max_rows_per_cloud, v.cols,
                                       _p(nn), _p(sim), _stream()), "morig_cosine_nn")
        return nn, sim

    def reserve_cus(self, n: int):
        """keep n CUs free of persistent EdgeConv workgroups (while FPS runs on a second stream)."""
        check(self.lib.morig_reserve_cus(int(n)), "morig_reserve_cus")

    # -- DeformNet glue (csrc/deform.hip) ----------------------------------------------------------------
    def sigmoid_minmax(self, x: Mat, ptr: torch.Tensor, n_meshes: int, out: Mat):
        """out = per-mesh min-max normalised sigmoid(x) (one column)."""
        _need_gpu(x.base, ptr, out.base)
        assert x.cols == 1 and out.cols == 1 and x.rows == out.rows and ptr.dtype == torch.int32
        check(self.lib.morig_sigmoid_minmax(x.ptr, x.ld, _p(ptr), n_meshes, out.ptr, out.ld, _stream()), "morig_sigmoid_minmax")

    def cosine_knn(self, y: Mat, ptr_y: torch.Tensor, x: Mat, ptr_x: torch.Tensor, n_clouds: int, max_rows_per_cloud: int,
                   k: int, vis: Mat = None, split: bool = False):
        """idx [y.rows, k] int32: the k most similar x rows of the same cloud per y row (-1 padded);
        split: rows with vis < 0.5 query rows with vis >= 0.5 of the same matrix."""
        _need_gpu(y.base, x.base, ptr_y, ptr_x)
        assert ptr_y.dtype == torch.int32 and ptr_x.dtype == torch.int32 and y.cols == x.cols
        idx = torch.empty((y.rows, k), dtype=torch.int32, device=y.base.device)
        if split:
            assert vis is not None and x.ptr == y.ptr
            ptr_x = ptr_y
        check(self.lib.morig_cosine_knn(y.ptr, y.ld, _p(ptr_y), x.ptr, x.ld, _p(ptr_x), n_clouds, max_rows_per_cloud, y.cols, k,
                                        vis.ptr if vis is not None else None, vis.ld if vis is not None else 0, int(split),
                                        _p(idx), _stream()), "morig_cosine_knn")
        return idx

    def flow_vote(self, mode: int, idx: torch.Tensor, feat_q: Mat, feat_s: Mat, pos_q, pos_s, vis: Mat, l1: Mat):
        """similarity-weighted voting into l1 = [flow(3) | vis]; mode 0: from points (pos_s - pos_q), all vertices;
        mode 1: invisible vertices from the flow of their visible neighbours."""
        _need_gpu(idx, feat_q.base, feat_s.base, vis.base, l1.base)
        assert idx.dtype == torch.int32 and idx.dim() == 2 and idx.shape[0] == feat_q.rows and l1.cols >= 4
        check(self.lib.morig_flow_vote(mode, _p(idx), idx.shape[1], feat_q.rows, feat_q.ptr, feat_q.ld, feat_s.ptr, feat_s.ld,
                                       feat_q.cols, pos_q.ptr if pos_q is not None else None, pos_q.ld if pos_q is not None else 0,
                                       pos_s.ptr if pos_s is not None else None, pos_s.ld if pos_s is not None else 0,
                                       vis.ptr, vis.ld, l1.ptr, l1.ld, _stream()), "morig_flow_vote")

    # -- joint extraction (csrc/joints.hip): float64 [n, 3] contiguous point sets -----------------------------
    @staticmethod
    def _pts64(pts: torch.Tensor):
        assert pts.dtype == torch.float64 and pts.dim() == 2 and pts.shape[1] == 3 and pts.is_contiguous()

    def inside_mask(self, pts: torch.Tensor, vox88: torch.Tensor, translate, scale: float, dims0: float) -> torch.Tensor:
        _need_gpu(pts, vox88)
        self._pts64(pts)
        assert vox88.dtype == torch.uint8 and vox88.numel() == 88 ** 3 and vox88.is_contiguous()
        keep = torch.empty(pts.shape[0], dtype=torch.uint8, device=pts.device)
        t = (C.c_double * 3)(*[float(v) for v in translate])
        check(self.lib.morig_inside_check(_p(pts), pts.shape[0], _p(vox88), t, float(scale), float(dims0), _p(keep), _stream()),
              "morig_inside_check")
        return keep.bool()

    def knn_bandwidth(self, pts: torch.Tensor, k: int) -> torch.Tensor:
        """device tensor [1] float64: mean distance to the k-th nearest neighbour (self included)."""
        _need_gpu(pts)
        self._pts64(pts)
        n = pts.shape[0]
        ws = torch.empty(n, dtype=torch.float64, device=pts.device)
        bw = torch.empty(1, dtype=torch.float64, device=pts.device)
        check(self.lib.morig_knn_bandwidth(_p(pts), n, k, _p(ws), _p(bw), _stream()), "morig_knn_bandwidth")
        return bw

    def meanshift(self, pts: torch.Tensor, weights: Optional[torch.Tensor], bandwidth: torch.Tensor, max_iter: int) -> torch.Tensor:
        _need_gpu(pts, bandwidth)
        self._pts64(pts)
        n = pts.shape[0]
        if weights is not None:
            assert weights.dtype == torch.float32 and weights.numel() == n and weights.is_contiguous()
        a, b = torch.empty_like(pts), torch.empty_like(pts)
        state = torch.empty(max(max_iter, 1), dtype=torch.float64, device=pts.device)
        in_a = C.c_int32(0)
        check(self.lib.morig_meanshift(_p(pts), _p(weights), n, _p(bandwidth), max_iter, _p(a), _p(b), _p(state), C.byref(in_a),
                                       _stream()), "morig_meanshift")
        return a if in_a.value else b

    def nms_counts(self, pts: torch.Tensor, bandwidth: torch.Tensor) -> torch.Tensor:
        _need_gpu(pts, bandwidth)
        self._pts64(pts)
        counts = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
        check(self.lib.morig_nms_counts(_p(pts), pts.shape[0], _p(bandwidth), _p(counts), _stream()), "morig_nms_counts")
        return counts

    def nms_greedy(self, pts: torch.Tensor, attn: torch.Tensor, bandwidth: torch.Tensor, order: torch.Tensor, thrd_density: float,
                   thrd_attn: float) -> torch.Tensor:
        _need_gpu(pts, attn, bandwidth, order)
        self._pts64(pts)
        n = pts.shape[0]
        assert attn.dtype == torch.float32 and attn.numel() == n and order.dtype == torch.int32 and order.numel() == n
        alive = torch.empty(n, dtype=torch.uint8, device=pts.device)
        check(self.lib.morig_nms_greedy(_p(pts), _p(attn), n, _p(bandwidth), _p(order), float(thrd_density), float(thrd_attn),
                                        _p(alive), _stream()), "morig_nms_greedy")
        return alive.bool()

    # -- the same stages over the point sets of several meshes (ptr: int32 [B + 1] row offsets on the device) ----------------
    def knn_bandwidth_batched(self, pts: torch.Tensor, ptr: torch.Tensor, max_n: int, quantile: float) -> torch.Tensor:
        _need_gpu(pts, ptr)
        self._pts64(pts)
        B = ptr.numel() - 1
        ws = torch.empty(pts.shape[0], dtype=torch.float64, device=pts.device)
        bw = torch.empty(B, dtype=torch.float64, device=pts.device)
        check(self.lib.morig_knn_bandwidth_batched(_p(pts), _p(ptr), B, pts.shape[0], max_n, float(quantile), _p(ws), _p(bw), _stream()),
              "morig_knn_bandwidth_batched")
        return bw

    def meanshift_batched(self, pts: torch.Tensor, weights: Optional[torch.Tensor], ptr: torch.Tensor, max_n: int,
                          bandwidth: torch.Tensor, max_iter: int) -> torch.Tensor:
        _need_gpu(pts, ptr, bandwidth)
        self._pts64(pts)
        B, n = ptr.numel() - 1, pts.shape[0]
        if weights is not None:
            assert weights.dtype == torch.float32 and weights.numel() == n and weights.is_contiguous()
        a, b = torch.empty_like(pts), torch.empty_like(pts)
        state = torch.empty(max(max_iter, 1) * B, dtype=torch.float64, device=pts.device)
        in_a = C.c_int32(0)
        check(self.lib.morig_meanshift_batched(_p(pts), _p(weights), _p(ptr), B, n, max_n, _p(bandwidth), max_iter, _p(a), _p(b),
                                               _p(state), C.byref(in_a), _stream()), "morig_meanshift_batched")
        return a if in_a.value else b

    def meanshift_batched_sorted(self, pts: torch.Tensor, weights: Optional[torch.Tensor], ptr: torch.Tensor, max_n: int,
                                 bandwidth: torch.Tensor, max_iter: int, with_counts: bool = False):
        """meanshift_batched on Morton-sorted points with bounding-box culling of source boxes (csrc/joints.hip); the result is
        returned in the caller's point order. Same sums up to the order of the additions (skipped pairs contribute exactly 0).
        with_counts: also the neighbour counts of the modes within the bandwidth (nms_counts_batched's integers), taken in the sorted
        order with the same culling -> (modes, counts)."""
        _need_gpu(pts, ptr, bandwidth)
        self._pts64(pts)
        B, n = ptr.numel() - 1, pts.shape[0]
        keys = torch.empty(n, dtype=torch.int64, device=pts.device)
        check(self.lib.morig_morton_keys(_p(pts), _p(ptr), B, n, _p(keys), _stream()), "morig_morton_keys")
        perm = torch.argsort(keys, stable=True)       # mirrored points on the x = 0 plane share a Morton key: a stable order keeps the fp64 summation order, and with it the modes' last bits, the same from run to run
        ps = pts[perm].contiguous()
        ws = None if weights is None else weights[perm].contiguous()
        a, b = torch.empty_like(ps), torch.empty_like(ps)
        state = torch.empty(max(max_iter, 1) * B, dtype=torch.float64, device=pts.device)
        bbox = torch.empty(2 * B * ((max_n + 31) // 32) * 6, dtype=torch.float64, device=pts.device)
        in_a = C.c_int32(0)
        check(self.lib.morig_meanshift_sorted(_p(ps), _p(ws), _p(ptr), B, n, max_n, _p(bandwidth), max_iter, _p(a), _p(b), _p(state),
                                              _p(bbox), C.byref(in_a), _stream()), "morig_meanshift_sorted")
        modes_sorted = a if in_a.value else b
        out = torch.empty_like(pts)
        out[perm] = modes_sorted
        if not with_counts:
            return out
        cs = torch.empty(n, dtype=torch.int32, device=pts.device)
        check(self.lib.morig_nms_counts_sorted(_p(modes_sorted), _p(ptr), B, n, max_n, _p(bandwidth), _p(bbox), _p(cs), _stream()),
              "morig_nms_counts_sorted")
        counts = torch.empty_like(cs)
        counts[perm] = cs
        return out, counts

    def nms_counts_batched(self, pts: torch.Tensor, ptr: torch.Tensor, max_n: int, bandwidth: torch.Tensor) -> torch.Tensor:
        _need_gpu(pts, ptr, bandwidth)
        self._pts64(pts)
        counts = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
        check(self.lib.morig_nms_counts_batched(_p(pts), _p(ptr), ptr.numel() - 1, pts.shape[0], max_n, _p(bandwidth), _p(counts),
                                                _stream()), "morig_nms_counts_batched")
        return counts

    def nms_greedy_batched(self, pts: torch.Tensor, attn: torch.Tensor, ptr: torch.Tensor, bandwidth: torch.Tensor,
                           order_local: torch.Tensor, thrd_density: float, thrd_attn: float) -> torch.Tensor:
        _need_gpu(pts, attn, ptr, bandwidth, order_local)
        self._pts64(pts)
        n = pts.shape[0]
        assert attn.dtype == torch.float32 and attn.numel() == n and order_local.dtype == torch.int32 and order_local.numel() == n
        alive = torch.empty(n, dtype=torch.uint8, device=pts.device)
        check(self.lib.morig_nms_greedy_batched(_p(pts), _p(attn), _p(ptr), ptr.numel() - 1, n, _p(bandwidth), _p(order_local),
                                                float(thrd_density), float(thrd_attn), _p(alive), _stream()), "morig_nms_greedy_batched")
        return alive.bool()

    def gather_rows(self, src: Mat, idx: torch.Tensor, dst: Mat):
        _need_gpu(src.base, idx, dst.base)
        assert idx.dtype == torch.int32 and dst.rows == idx.numel() and dst.cols == src.cols
        check(self.lib.morig_gather_rows(src.ptr, src.ld, _p(idx), idx.numel(), src.cols, dst.ptr, dst.ld, _stream()),
              "morig_gather_rows")

    # -- small ops ------------------------------------------------------------------------------------
    def copy2d(self, src: Mat, dst: Mat):
        _need_gpu(src.base, dst.base)
        assert src.rows == dst.rows and src.cols == dst.cols
        check(self.lib.morig_copy2d(src.ptr, src.ld, dst.ptr, dst.ld, src.rows, src.cols, _stream()), "morig_copy2d")

    def copy2d_pad(self, src: Mat, dst: Mat, split: bool = False):
        """dst window (wider than src) = [src | zeros]; split: written in the split-fp16 activation layout."""
        _need_gpu(src.base, dst.base)
        assert src.rows == dst.rows and dst.cols >= src.cols
        if split and not self.fast:
            raise MorigNativeError("split-fp16 activations requested on the fp32 path (plan bug)")
        check(self.lib.morig_copy2d_pad(src.ptr, src.ld, src.rows, src.cols, dst.ptr, dst.ld, dst.cols, int(split),
                                        _p(self._flag(src.base.device)), _stream()), "morig_copy2d_pad")

    def copy2d_rep(self, src: Mat, dst: Mat, replicas: int, dst_row_step: int, src_col_step: int = 0, split: bool = False):
        """`replicas` copies in one launch: copy r = [src window shifted r * src_col_step columns | zeros] into the dst
        window shifted r * dst_row_step rows (dst describes replica 0)."""
        _need_gpu(src.base, dst.base)
        assert src.rows == dst.rows and dst.cols >= src.cols and replicas >= 1
        assert dst.row0 + dst.rows + (replicas - 1) * dst_row_step <= dst.base.shape[0]
        assert src.col0 + src.cols + (replicas - 1) * src_col_step <= src.base.shape[1]
        if split and not self.fast:
            raise MorigNativeError("split-fp16 activations requested on the fp32 path (plan bug)")
        check(self.lib.morig_copy2d_pad_rep(src.ptr, src.ld, src.rows, src.cols, src_col_step, dst.ptr, dst.ld, dst.cols, replicas,
                                            dst_row_step, int(split), _p(self._flag(src.base.device)), _stream()), "morig_copy2d_pad_rep")

    def gather_cols(self, src: Mat, cols: torch.Tensor, dst: Mat):
        _need_gpu(src.base, cols, dst.base)
        assert cols.dtype == torch.int32 and dst.cols == cols.numel() and src.rows == dst.rows
        check(self.lib.morig_gather_cols(src.ptr, src.ld, _p(cols), cols.numel(), dst.ptr, dst.ld, src.rows, _stream()),
              "morig_gather_cols")

    def make_seg(self, batch: torch.Tensor, n_graphs: int, replicas: int) -> torch.Tensor:
        _need_gpu(batch)
        b = batch if (batch.dtype == torch.int64 and batch.is_contiguous()) else batch.long().contiguous()
        n = b.numel()
        seg = torch.empty(n * replicas, dtype=torch.int32, device=b.device)
        check(self.lib.morig_make_seg(_p(b), n, n_graphs, replicas, _p(seg), _stream()), "morig_make_seg")
        return seg

    def rownorm(self, x: Mat, rows_per_rep: int, replicas: int, y: torch.Tensor, ld_row: int, ld_rep: int):
        _need_gpu(x.base, y)
        assert x.rows == rows_per_rep * replicas
        check(self.lib.morig_rownorm(x.ptr, x.ld, rows_per_rep, replicas, x.cols, _p(y), ld_row, ld_rep, _stream()),
              "morig_rownorm")

    def cls_attention(self, x: torch.Tensor, g: torch.Tensor, cls: torch.Tensor, y: Mat):
        _need_gpu(x, g, cls, y.base)
        n, T, Cc = x.shape
        assert x.is_contiguous() and g.is_contiguous() and cls.is_contiguous()
        heads = g.shape[0]
        check(self.lib.morig_cls_attention(_p(x), n, T, Cc, heads, _p(g), _p(cls), y.ptr, y.ld, _stream()),
              "morig_cls_attention")

    def frame_reduce(self, x: torch.Tensor, mode: str, y: Mat):
        _need_gpu(x, y.base)
        n, T, Cc = x.shape
        assert x.is_contiguous()
        check(self.lib.morig_frame_reduce(_p(x), n, T, Cc, 0 if mode == "mean" else 1, y.ptr, y.ld, _stream()),
              "morig_frame_reduce")


_ops: Optional[NativeOps] = None


def get_ops() -> NativeOps:
    """The process-wide op layer. Raises (loudly) when the HIP library or a GPU is missing."""
    global _ops
    if _ops is None:
        _ops = NativeOps()
    return _ops


# ---------------------------------------------------------------------------------------------
# profiling helpers for bench.py
# ---------------------------------------------------------------------------------------------
def prof_enable(on: bool) -> None:
    load_library().morig_prof_enable(1 if on else 0)


def prof_reset() -> None:
    load_library().morig_prof_reset()


def prof_collect():
    """-> {kernel name: dict(launches, ms, flops, bytes)} for kinds that ran."""
    lib = load_library()
    out = {}
    k = 0
    while True:
        nm = lib.morig_prof_name(k)
        if nm is None:
            break
        n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        check(lib.morig_prof_collect(k, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), "morig_prof_collect")
        if n.value:
            sym = lib.morig_prof_symbol(k)
            out[nm.decode()] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value,
                                    symbol=sym.decode() if sym else None)
        k += 1
    return out


def device_info():
    lib = load_library()
    cu, lds, clk = C.c_int(), C.c_int(), C.c_int()
    arch = C.create_string_buffer(64)
    st = lib.morig_device_info(C.byref(cu), C.byref(lds), C.byref(clk), arch, 64)
    return dict(status=st, cu_count=cu.value, lds_bytes_per_cu=lds.value, clock_khz=clk.value, arch=arch.value.decode())
