"""ctypes binding of libpvs_egnn.so (C ABI declared in include/pvs_egnn.h)."""
import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get('PVS_EGNN_LIB', _HERE / 'libpvs_egnn.so'))

# PVS_* layer flags (pvs_egnn.h)
RESIDUAL, EDGE_RESIDUAL, EDGE_ATTENTION, NORMALIZE = 1 << 0, 1 << 1, 1 << 2, 1 << 3
TANH, GRAPHNORM, UPDATE_COORDS, PERM_INVARIANT = 1 << 4, 1 << 5, 1 << 6, 1 << 7
NODE_ATTENTION, GATED_RESIDUAL, REZERO, SOFTMAX_ATT = 1 << 8, 1 << 9, 1 << 10, 1 << 11
ACT_CODES = {'sigmoid': 0, 'tanh': 1, 'relu': 2, 'silu': 3, 'identity': 4}

PARAM_FIELDS = (
    'edge_w1', 'edge_b1', 'edge_w2', 'edge_b2', 'coord_w1', 'coord_b1', 'coord_w2', 'att_w',
    'att_b', 'node_w1', 'node_b1', 'node_w2', 'node_b2', 'gn_weight', 'gn_bias', 'gn_mean_scale',
    'node_att_w', 'node_att_b', 'edge_gate', 'node_gate')


class PvsLayerDesc(C.Structure):
    _fields_ = [('hidden', C.c_int32), ('n_edge_attr', C.c_int32), ('flags', C.c_uint32),
                ('att_act', C.c_int32)]


class PvsGraph(C.Structure):
    _fields_ = [('n_nodes', C.c_int32), ('n_edges', C.c_int32), ('rowptr', C.c_void_p),
                ('row', C.c_void_p), ('col', C.c_void_p), ('etype', C.c_void_p),
                ('perm', C.c_void_p), ('colptr', C.c_void_p), ('cedge', C.c_void_p),
                ('inv_deg', C.c_void_p), ('n_edges_dev', C.c_void_p), ('graph_eptr', C.c_void_p),
                ('n_graphs', C.c_int32)]


class PvsLayerParams(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in PARAM_FIELDS]


class PvsLayerGrads(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in PARAM_FIELDS]


class PvsStackStrides(C.Structure):
    _fields_ = [(name, C.c_int64) for name in ('h_mid', 'x_mid', 'att', 'node_att', 'saved')]


_PROTOTYPES = {
    'pvs_last_error': (C.c_char_p, []),
    'pvs_version': (C.c_int, []),
    'pvs_graph_prepare_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32]),
    'pvs_graph_prepare': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] +
                          [C.c_void_p] * 9 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_graph_prepare_runs_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'pvs_graph_prepare_runs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32] +
                               [C.c_void_p] * 11 + [C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_radius_graph_state_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'pvs_radius_graph_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'pvs_radius_graph_count': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_radius_graph_fill': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32] +
                              [C.c_void_p] * 10 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_graph_min_label_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    'pvs_dropout_adj_workspace_bytes': (C.c_size_t, [C.c_int32]),
    'pvs_dropout_adj_mark': (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_dropout_adj_fill': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    'pvs_rows_to_input_order': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_void_p]),
    'pvs_rows_to_sorted_order': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_void_p]),
    'pvs_egnn_layer_saved_floats': (C.c_size_t, [C.POINTER(PvsLayerDesc), C.c_int32, C.c_int32]),
    'pvs_egnn_layer_workspace_bytes': (C.c_size_t, [C.POINTER(PvsLayerDesc), C.c_int32, C.c_int32,
                                                    C.c_int32]),
    'pvs_egnn_layer_fwd': (C.c_int, [C.POINTER(PvsLayerDesc), C.POINTER(PvsGraph),
                                     C.POINTER(PvsLayerParams)] + [C.c_void_p] * 9 +
                           [C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_screen_graph_state_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'pvs_screen_graph_build': (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_double, C.c_double, C.c_int32,
                                                                              C.c_int32] + [C.c_void_p] * 10 +
                               [C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_graph_filter_workspace_bytes': (C.c_size_t, [C.c_int32]),
    'pvs_graph_filter_ligand_edges': (C.c_int, [C.POINTER(PvsGraph), C.c_void_p, C.c_int32] + [C.c_void_p] * 5 +
                                      [C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_egnn_layer_edge_sums': (C.c_int, [C.POINTER(PvsLayerDesc), C.POINTER(PvsGraph),
                                           C.POINTER(PvsLayerParams)] + [C.c_void_p] * 4 +
                                 [C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_egnn_layer_fwd_partial': (C.c_int, [C.POINTER(PvsLayerDesc), C.POINTER(PvsGraph),
                                             C.POINTER(PvsLayerParams)] + [C.c_void_p] * 9 +
                                   [C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_egnn_layer_bwd': (C.c_int, [C.POINTER(PvsLayerDesc), C.POINTER(PvsGraph),
                                     C.POINTER(PvsLayerParams)] + [C.c_void_p] * 11 +
                           [C.POINTER(PvsLayerGrads), C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_egnn_stack_workspace_bytes': (C.c_size_t, [C.POINTER(PvsLayerDesc), C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'pvs_egnn_stack_fwd': (C.c_int, [C.POINTER(PvsLayerDesc), C.POINTER(PvsLayerParams), C.c_int32, C.POINTER(PvsGraph),
                                     C.POINTER(PvsStackStrides)] + [C.c_void_p] * 9 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_egnn_stack_bwd': (C.c_int, [C.POINTER(PvsLayerDesc), C.POINTER(PvsLayerParams), C.c_int32, C.POINTER(PvsGraph),
                                     C.POINTER(PvsStackStrides)] + [C.c_void_p] * 10 +
                           [C.POINTER(PvsLayerGrads), C.c_void_p, C.c_size_t, C.c_void_p]),
    'pvs_linear_fwd': (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_void_p]),
    'pvs_linear_bwd_workspace_bytes': (C.c_size_t, [C.c_int32] * 3),
    'pvs_linear_bwd': (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_void_p, C.c_size_t,
                                                                      C.c_void_p]),
    'pvs_mean_pool_fwd': (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 2 + [C.c_void_p]),
    'pvs_mean_pool_bwd': (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p]),
    'pvs_pool_head_fwd': (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_void_p]),
    'pvs_pool_head_bwd': (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 4 + [C.c_void_p]),
    'pvs_bce_logits_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pvs_scale_by_device_scalar': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'pvs_segment_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32]),
    'pvs_segment_reduce_fwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_void_p]),
    'pvs_segment_reduce_bwd': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'pvs_adam_clip_step': (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_float, C.c_float,
                                     C.c_double, C.c_double, C.c_float, C.c_void_p]),
    'pvs_adam_clip_step_dev': (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_float, C.c_float,
                                         C.c_void_p, C.c_float, C.c_void_p]),
    'pvs_profile_enable': (C.c_int, [C.c_int]),
    'pvs_profile_reset': (C.c_int, []),
    'pvs_profile_read': (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'pvs_profile_read_each': (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64)]),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)
_lib = None


def lib():
    """The loaded library. Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; '
                f'g.build()"` (or `make -C pointvs_amd/csrc`). pointvs_amd has no CPU fallback.')
        # torch FIRST: it brings its own copy of the HIP runtime (SONAME libamdhip64.so.7, like the system's one this
        # library is linked against). Loaded before torch, the library would bind the system runtime and torch its
        # bundled one - two runtimes in one process, and the streams / device state of one are nothing to the other
        # ("no ROCm-capable device is detected" from the first launch; seen when build() and smoke() ran in one process).
        import torch  # noqa: F401
        handle = C.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in _PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = restype, argtypes
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().pvs_last_error().decode(errors='replace')
        raise RuntimeError(f'{what} failed ({rc}): {msg}')


def ptr(t):
    """Device pointer of a tensor (None -> NULL). The tensor must be contiguous."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError('non-contiguous tensor handed to the C ABI')
    return t.data_ptr()


def require_hip(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                'pointvs_amd kernels run on a HIP device only (got a CPU tensor); there is no CPU '
                'path in the product - the CPU oracle lives in oracle/ and is test-only')


def stream(dev):
    """Raw handle (hipStream_t as an int) of torch's CURRENT stream on `dev`, as every C-ABI call takes it. The raw
    accessor, not torch.cuda.current_stream(dev).cuda_stream: that builds a Stream object (5 us a call on the host, a
    dozen calls per training step of a small batch)."""
    import torch
    idx = dev.index
    if idx is None:
        idx = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)

