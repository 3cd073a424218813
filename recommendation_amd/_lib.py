"""ctypes binding of libmi355rec.so (the C ABI declared in include/mi355rec.h).

There is no CPU fallback: if the shared library is missing the import of any
op raises, so a GPU run can never silently pass on something else.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355REC_LIB: another build of the same library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get('MI355REC_LIB') or os.path.join(_HERE, 'libmi355rec.so')

MI_CHECK_RANGE, MI_CHECK_INCREASING, MI_CHECK_NONNEG = 1, 2, 4
MI_IDERR_RANGE, MI_IDERR_ORDER, MI_IDERR_INTERNAL, MI_IDERR_CAPACITY = 1, 2, 4, 8
MI_POOL_SUM, MI_POOL_MEAN = 0, 1
MI_DT_F32, MI_DT_BF16 = 0, 1
MI_OPT_SGD, MI_OPT_ADAGRAD_ROWWISE, MI_OPT_ADAM_TF = 0, 1, 2
MI_NOALLOC_MAX_ENTRIES = 1 << 21
MI_COMM_ID_BYTES = 128


class MiError(RuntimeError):
    """A C-ABI call returned a negative MI_ERR_* code."""


class BwdPlan(Structure):
    """struct mi_bwd_plan"""
    _fields_ = [
        ('sorted_rows', c_void_p), ('perm', c_void_p), ('unique_rows', c_void_p),
        ('seg_offsets', c_void_p), ('counts', c_void_p),
        ('n_entries', c_int64), ('n_bags', c_int64), ('total_rows', c_int64), ('offsets', c_void_p),
        ('bag_len', c_int), ('T', c_int), ('pool_mode', c_int), ('long_list', c_void_p),
        ('indices', c_void_p), ('idx_bits', c_int), ('table_offsets', c_void_p),
        ('bag_scale', c_void_p), ('entry_bag', c_void_p),
    ]


class IdColumn(Structure):
    """struct mi_id_column"""
    _fields_ = [('ids', c_void_p), ('n_bags', c_int64), ('rows', c_int64), ('bag_len', c_int32),
                ('check_flags', c_int32)]


class FmTable(Structure):
    """struct mi_fm_table"""
    _fields_ = [('rec', c_void_p), ('stride', c_int64), ('dim', c_int), ('off_w1', c_int), ('off_vs0', c_int),
                ('off_vs1', c_int), ('off_ws0', c_int), ('off_ws1', c_int)]


class AdamParam(Structure):
    """mi_adam_param"""
    _fields_ = [('w', c_void_p), ('m', c_void_p), ('v', c_void_p), ('w_bf16', c_void_p), ('w_bf16_t', c_void_p),
                ('grad_out', c_void_p)]


CROSS_MAX_LAYERS = 8


class CrossNet(Structure):
    """mi_cross_net"""
    _fields_ = [('n_layers', c_int), ('width', c_int), ('W', AdamParam * CROSS_MAX_LAYERS),
                ('b', AdamParam * CROSS_MAX_LAYERS)]


class DinLau(Structure):
    """mi_din_lau"""
    _fields_ = [('dim', c_int), ('h1', c_int), ('h2', c_int), ('L', c_int), ('factored', c_int), ('W1', AdamParam),
                ('b1', AdamParam), ('W2', AdamParam), ('b2', AdamParam), ('w3', AdamParam), ('W1f', c_void_p),
                ('W1fT', c_void_p), ('Wq', c_void_p), ('WqT', c_void_p)]


class GemmEpilogue(Structure):
    """struct mi_gemm_epilogue"""
    _fields_ = [('mode', c_int), ('flags', c_int), ('out', c_void_p), ('ld_out', c_int64), ('bias', c_void_p),
                ('in0', c_void_p), ('ld_in0', c_int64), ('in1', c_void_p), ('ld_in1', c_int64),
                ('out1', c_void_p), ('ld_out1', c_int64), ('bias_ld', c_int64), ('bias_rows', c_int)]


P = c_void_p
_SIGNATURES = {
    'mi_version': (c_char_p, []),
    'mi_last_error': (c_char_p, []),
    'mi_embag_fwd': (c_int, [P, P, c_int, c_int, P, c_int, P, c_int, c_int64, c_int, c_int, P, P, c_int,
                             c_int64, P, P]),
    'mi_onehot_dense': (c_int, [P, c_int, c_int, c_int64, c_int, c_int, P, P, P]),
    'mi_validate_ids': (c_int, [P, c_int, P, c_int, P, c_int, c_int64, c_int, P, P]),
    'mi_validate_id_columns': (c_int, [P, c_int, c_int, P, P]),
    'mi_concat_id_columns': (c_int, [P, c_int, c_int, c_int64, P, P]),
    'mi_stage_from_host': (c_int, [P, P, c_size_t, P]),
    'mi_embag_bag_info': (c_int, [P, c_int, P, c_int, P, c_int, c_int64, P, P, P]),
    'mi_embag_bwd_workspace_bytes': (c_size_t, [c_int64]),
    'mi_embag_bwd_prepare': (c_int, [P, c_int, c_int64, P, c_int, P, c_int, c_int64, c_int64, P, P, P, P, P,
                                     P, c_size_t, P]),
    'mi_embag_bwd_prepare_noalloc': (c_int, [P, c_int, c_int64, P, c_int, P, c_int, c_int64, c_int64, P, P, P, P, P,
                                             P]),
    'mi_sort_pairs_u32': (c_int, [P, P, c_int64, c_int, P, c_size_t, P]),
    'mi_embag_bwd_dense_grad': (c_int, [POINTER(BwdPlan), P, c_int64, c_int, c_int64, P, P]),
    'mi_embag_bwd_apply_sgd': (c_int, [POINTER(BwdPlan), P, c_int64, c_int, P, c_float, P]),
    'mi_embag_bwd_apply_adagrad_rowwise': (c_int, [POINTER(BwdPlan), P, c_int64, c_int, P, P, c_float, c_float, P]),
    'mi_embag_bwd_apply_adam_tf': (c_int, [POINTER(BwdPlan), P, c_int64, c_int, P, P, P, c_float, c_float,
                                           c_float, c_float, P]),
    'mi_embag_bwd_apply_adam_dense_tf': (c_int, [POINTER(BwdPlan), P, c_int64, c_int, c_int64, P, P, P, c_float,
                                                 c_float, c_float, c_float, P]),
    'mi_linear_fwd': (c_int, [P, c_int64, P, P, P, c_int64, c_int64, c_int, c_int, P]),
    'mi_linear_bwd_workspace_bytes': (c_size_t, [c_int64, c_int, c_int]),
    'mi_linear_bwd': (c_int, [P, c_int64, P, P, c_int64, P, c_int64, P, P, c_int64, c_int, c_int, P, c_size_t, P]),
    'mi_softmax_heads': (c_int, [P, c_int64, c_int64, c_int, c_int, P, P, c_float, P, P, P, P, P, P, P]),
    'mi_hit_rows': (c_int, [P, P, c_int64, c_int, P, P]),
    'mi_colsum': (c_int, [P, c_int64, c_int64, c_int, P, P]),
    'mi_reduce_sum': (c_int, [P, c_int64, c_float, P, P]),
    'mi_adam_tf_dense': (c_int, [P, P, P, P, c_int64, c_float, P, c_float, c_float, c_float, P]),
    'mi_fm_fwd': (c_int, [POINTER(FmTable), P, c_int, P, c_int, c_int64, P, c_int, P, P, P, c_float, P, P, P, P, P,
                          P]),
    'mi_fm_bwd_apply': (c_int, [POINTER(BwdPlan), POINTER(FmTable), P, P, P, c_int, c_float, P, c_float, c_float,
                                c_float, P]),
    'mi_fm_bwd_dense': (c_int, [P, P, c_int, c_int64, P, P]),
    'mi_fm_dense_step': (c_int, [P, P, c_int, c_int64, P, P, P, P, c_float, P, c_float, c_float, c_float, P, P]),
    'mi_adam_schedule_step': (c_int, [P, c_float, c_float, c_float, P]),
    'mi_gemm_bf16': (c_int, [c_int, c_int, c_int64, c_int64, c_int64, P, c_int64, P, c_int64, c_int,
                             POINTER(GemmEpilogue), P]),
    'mi_pack_dense_bf16': (c_int, [P, c_int, c_int64, P, c_int64, c_int, c_int, P]),
    'mi_colsum_bf16': (c_int, [P, c_int64, c_int64, c_int, c_int, P, c_int64, P]),
    'mi_slab_adam': (c_int, [P, c_int, c_int64, c_int64, P, P, P, c_float, P, c_float, c_float, c_float, P, P, P]),
    'mi_head_fwd': (c_int, [P, c_int64, c_int, P, P, P, c_float, c_int64, P, P, P, P]),
    'mi_head_bwd': (c_int, [P, P, P, c_int64, c_int, c_int64, P, c_int64, c_int, P, c_int64, P]),
    'mi_cast_bf16': (c_int, [P, c_int64, P, P]),
    'mi_transpose_bf16': (c_int, [P, c_int64, c_int, c_int, P, c_int64, P]),
    'mi_cross_input_grad': (c_int, [P, P, c_int, c_int64, c_int, c_int64, P, c_int64, P]),
    'mi_din_lau_build': (c_int, [P, c_int64, c_int, P, P, c_int, P, c_int64, c_int, P, P, P, c_int64, c_int, P, P]),
    'mi_din_lau_pool': (c_int, [P, c_int, P, P, c_int64, c_int, c_int, P, c_int64, P]),
    'mi_din_lau_pool_bwd': (c_int, [P, c_int, P, c_int64, P, c_int64, c_int, c_int, P, P]),
    'mi_din_lau_scatter_fact': (c_int, [P, P, P, P, c_int64, P, c_int64, P, c_int64, P, c_int64, c_int, c_int, P, P]),
    'mi_din_qterm_part_floats': (c_int64, [c_int64, c_int, c_int]),
    'mi_din_qterm_fwd': (c_int, [P, c_int64, P, P, c_int64, c_int, c_int, P, P]),
    'mi_din_qterm_bwd': (c_int, [P, c_int64, P, P, c_int64, c_int, c_int, P, c_int64, P, c_int64, P, P]),
    'mi_din_w1_step': (c_int, [P, c_int, c_int64, P, c_int, c_int64, c_int, c_int, P, P, P, c_float, P, c_float, c_float,
                               c_float, P, P, P, P, P, P]),
    'mi_din_lau_scatter': (c_int, [P, P, P, P, c_int64, P, c_int64, P, c_int64, c_int, c_int, P, P]),
    'mi_bce_logits': (c_int, [P, P, c_int64, c_float, P, P, P]),
    'mi_mul_bf16': (c_int, [P, c_int64, P, c_int64, c_int64, c_int, P, c_int64, P]),
    'mi_gemm_wgrad_splits': (c_int, [c_int64, c_int64, c_int64]),
    'mi_cross_workspace_bytes': (c_size_t, [c_int64, c_int, c_int]),
    'mi_cross_workspace_part': (c_int, [c_int64, c_int, c_int, c_int, c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    'mi_cross_fwd': (c_int, [POINTER(CrossNet), P, c_int64, c_int64, P, c_size_t, POINTER(c_void_p), P]),
    'mi_cross_bwd': (c_int, [POINTER(CrossNet), P, c_int64, c_int64, P, c_size_t, c_int, c_int, c_float, P, c_float,
                             c_float, c_float, P, c_int64, P]),
    'mi_din_lau_workspace_bytes': (c_size_t, [c_int64, POINTER(DinLau)]),
    'mi_din_lau_workspace_part': (c_int, [c_int64, POINTER(DinLau), c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    'mi_din_lau_fwd': (c_int, [POINTER(DinLau), P, c_int64, P, P, c_int, P, c_int64, P, c_size_t, P, c_int64, P, P, P]),
    'mi_din_lau_fwd_fused_ok': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'mi_din_lau_fwd_fused': (c_int, [P, c_int64, P, P, c_int, P, c_int64, c_int, P, P, P, P, P, P, P, P, P, P, P, P, c_int64, P, P, P]),
    'mi_din_lau_bwd_part_floats': (c_int, []),
    'mi_din_lau_bwd_fused': (c_int, [c_int64, c_int, P, P, P, P, P, P, c_int64, P, c_int64, P, P, P, P, P, P, P, P, c_int, P, P]),
    'mi_din_lau_bwd': (c_int, [POINTER(DinLau), c_int64, P, c_size_t, P, c_int64, P, c_int64, P, c_float, P, c_float,
                               c_float, c_float, P, P]),
    'mi_fm_fwd_sort_max_batch': (c_int, []),
    'mi_fm_lds_sort_single_max': (c_int, []),
    'mi_fm_fwd_sort': (c_int, [POINTER(FmTable), P, c_int, P, c_int, c_int64, P, c_int, P, P, P, c_float, P, P, P, P,
                               P, c_int64, P, P, P, c_float, c_float, c_float, c_int, P]),
    'mi_fm_sort_ids': (c_int, [P, c_int, P, c_int, c_int64, c_int, c_int64, P, P, P]),
    'mi_fm_bwd_apply_dense': (c_int, [POINTER(BwdPlan), POINTER(FmTable), P, P, c_int, c_float, P, c_float, c_float,
                                      c_float, P, c_int, P, P, P, P, c_float, P, P, P]),
    'mi_fm_bwd_apply_dense_stage': (c_int, [POINTER(BwdPlan), POINTER(FmTable), P, P, c_int, c_float, P, c_float, c_float,
                                            c_float, P, c_int, P, P, P, P, c_float, P, P, P, P, c_size_t, P]),
    'mi_parse_records': (c_int, [c_char_p, c_int64, c_int, c_int, c_int64, P, P, POINTER(c_int64), POINTER(c_int64)]),
    'mi_shard_block_ints': (c_int64, [c_int64, c_int]),
    'mi_shard_route': (c_int, [P, c_int, P, c_int, c_int64, c_int, P, P, P]),
    'mi_shard_owner_fwd_sort': (c_int, [POINTER(FmTable), c_int64, P, c_int, c_int, c_int, P, c_int64, c_int64, P, P, P, P, P,
                                        c_float, c_float, c_float, P, P]),
    'mi_fm_combine': (c_int, [P, c_int, c_int64, c_int, P, c_int, P, P, P, c_float, P, P, c_int, P, P]),
    'mi_shard_dense_part_floats': (c_int64, [c_int64, c_int, c_int]),
    'mi_shard_bwd_scratch_floats': (c_int64, [c_int64]),
    'mi_shard_owner_bwd_dense': (c_int, [POINTER(FmTable), c_int64, P, P, P, c_int64, P, c_int, c_int, c_int64, c_int64, c_int, c_float,
                                         P, c_float, c_float, c_float, c_int, P, P, P, c_float, P, P, P, P]),
    'mi_shard_bag_slots_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'mi_shard_bag_bucket_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'mi_shard_bag_bucket': (c_int, [P, c_int, P, c_int, c_int, c_int64, c_int, c_int, c_int64, c_int64, P, P, P, P, P,
                                    c_size_t, P, P]),
    'mi_shard_bag_slots': (c_int, [P, c_int, P, c_int, c_int, c_int64, c_int, c_int, c_int, c_int64, P, P, P, P, P,
                                   c_size_t, P, P]),
    'mi_shard_bag_gather': (c_int, [P, P, c_int, c_int, P, c_int, c_int64, c_int, c_int64, P, P, P, P]),
    'mi_shard_bag_combine': (c_int, [P, P, c_int, c_int64, c_int, c_int, P, c_int64, P]),
    'mi_shard_bag_pack': (c_int, [P, c_int64, c_int, c_int, P, P, c_int, c_int64, P, P]),
    'mi_embag_bwd_apply_rows': (c_int, [POINTER(BwdPlan), P, P, c_int, c_int, P, P, c_float, c_float, P]),
    'mi_comm_unique_id': (c_int, [P]),
    'mi_comm_init': (c_int, [P, c_int, c_int, POINTER(c_void_p)]),
    'mi_comm_destroy': (c_int, [P]),
    'mi_comm_size': (c_int, [P, POINTER(c_int), POINTER(c_int)]),
    'mi_a2a_indices': (c_int, [P, P, P, c_int64, P]),
    'mi_a2a_pooled': (c_int, [P, P, P, c_int64, P]),
    'mi_allgather_bytes': (c_int, [P, P, P, c_int64, P]),
    'mi_allreduce_dense': (c_int, [P, P, c_int64, P]),
    'mi_fm_dense_step_rows': (c_int, [P, c_int, c_int, c_int, c_int64, P, P, P, c_float, P, c_float, c_float,
                                      c_float, P, P]),
}

EXPORTS = tuple(sorted(_SIGNATURES))

_lib = None


def load():
    """Load the library once; raise with a build hint when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MiError(
            "%s not found: build it with `make -C %s` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the HIP path." % (LIB_PATH, os.path.join(_HERE, 'csrc')))
    # torch ships its own copy of the HIP runtime.  If this library pulled in the system's copy first, a process that
    # imports torch afterwards holds two runtimes and this library's launches fail with "no ROCm-capable device":
    # with torch loaded first both bind to one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header and library out of step
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning entry point and raise MiError on a negative code."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MiError("%s failed (%d): %s" % (name, rc, lib.mi_last_error().decode()))
    return rc
