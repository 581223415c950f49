"""Graph containers and the once-per-batch CSR/CSC preparation.

`Data` / `Batch` are minimal duck types of the PyG objects the reference hands to the path
(`.x .edge_index .edge_attr .pos .batch .y .lig_fname .rec_fname`,
/root/reference/point_vs/preprocessing/data_loaders.py:381-391): PyG is not assumed present.
`PreparedGraph` is what `pvs_graph_prepare` (include/pvs_egnn.h) produces from the int64 COO +
int64 one-hot; every layer's forward and backward of a step share one instance.
"""
import collections
import ctypes as C
import os

import torch

from . import _lib


class Data:
    """One graph: attribute bag, `.to(device)` moves tensors (PyG `Data` duck type)."""

    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)

    def to(self, device, non_blocking=False):
        for key, val in list(self.__dict__.items()):
            if torch.is_tensor(val):
                self.__dict__[key] = val.to(device, non_blocking=non_blocking)
        return self

    def keys(self):
        return list(self.__dict__.keys())


class Batch(Data):
    """Disjoint union of graphs with node-offset edge indices and a `batch` vector."""

    @staticmethod
    def from_data_list(items):
        merged = {}
        offset, shifted, batch_vec, ptr = 0, [], [], [0]
        for gid, item in enumerate(items):
            n = item.x.size(0)
            batch_vec.append(torch.full((n,), gid, dtype=torch.long))
            shifted.append(item.edge_index + offset)
            offset += n
            ptr.append(offset)
        for key in items[0].__dict__:
            vals = [getattr(item, key) for item in items]
            if key == 'edge_index':
                merged[key] = torch.cat(shifted, dim=1)
            elif torch.is_tensor(vals[0]):
                merged[key] = torch.cat([v.reshape(1) if v.dim() == 0 else v for v in vals], dim=0)
            elif vals[0] is None:
                merged[key] = None
            else:
                merged[key] = vals
        # 'generate_edges' = every graph's edge list is the reference loader's: the inter-molecular block then
        # the intra-molecular block, each row-major (preprocessing.py:108-142). Collation keeps that per graph,
        # so the batch can be prepared by merging sorted runs (pvs_graph_prepare_runs) instead of sorting.
        tags = {getattr(item, 'edge_layout', None) for item in items}
        merged['edge_layout'] = tags.pop() if len(tags) == 1 else None
        merged['batch'] = torch.cat(batch_vec)
        merged['ptr'] = torch.tensor(ptr, dtype=torch.long)
        merged['num_graphs'] = len(items)
        # host-side per-graph sizes (plain lists survive .to(device)): sizes without a device->host sync
        # (radius_graph's mask buffers, the screening batcher)
        merged['graph_node_counts'] = [int(item.x.size(0)) for item in items]
        merged['graph_edge_counts'] = [int(item.edge_index.size(1)) for item in items]
        return Batch(**merged)


class PreparedGraph:
    """Device-resident CSR (by row = edge_index[0]) + CSC (by col) of one batch."""

    def __init__(self, n_nodes, n_edges, n_edge_attr, tensors):
        self.n_nodes, self.n_edges, self.n_edge_attr = n_nodes, n_edges, n_edge_attr
        self.t = tensors  # keeps the storage alive for the struct's raw pointers
        g = _lib.PvsGraph()
        g.n_nodes, g.n_edges = n_nodes, n_edges
        for name in ('rowptr', 'row', 'col', 'etype', 'perm', 'colptr', 'cedge', 'inv_deg'):
            setattr(g, name, _lib.ptr(tensors.get(name)))
        self.c = g
        self._status_checked = False
        self._status_host = None
        self._status_event = None

    @property
    def perm(self):
        return self.t['perm']

    def set_graph_ptr(self, node_ptr):
        """node_ptr: device int tensor [B + 1] of the batch's node offsets (PyG `ptr`). Records where each
        graph's edges start in the sorted list (PvsGraph.graph_eptr): the fp16-split edge backward ends its
        tiles there. One small device gather, no host sync; optional (a batch of one graph needs none; a batch
        prepared from its 'generate_edges' layout has the table already: prepare_graph)."""
        if node_ptr is None or node_ptr.numel() <= 2 or self.c.graph_eptr:
            return
        idx = getattr(node_ptr, '_pvs_as_long', None)       # (cached on the tensor: a conversion launch per step otherwise)
        if idx is None or idx.device != self.t['rowptr'].device:
            idx = node_ptr.to(device=self.t['rowptr'].device, dtype=torch.long)
            try:
                node_ptr._pvs_as_long = idx
            except AttributeError:
                pass
        eptr = self.t['rowptr'].index_select(0, idx).contiguous()
        self.t['graph_eptr'] = eptr
        self.c.graph_eptr = _lib.ptr(eptr)
        self.c.n_graphs = int(eptr.numel()) - 1

    def poll_status(self):
        """Asynchronous validation: the first call queues a D2H copy of the status word, later
        calls (or check_status) raise once it has landed. Never blocks the stream."""
        if self._status_checked or torch.cuda.is_current_stream_capturing():
            return   # (no host-visible validation inside a captured step)
        if self._status_event is None:
            self._status_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            self._status_host.copy_(self.t['status'], non_blocking=True)
            self._status_event = torch.cuda.Event()
            self._status_event.record(torch.cuda.current_stream(self.t['status'].device))
            _PENDING.append(self)
        for pg in list(_PENDING):
            if pg._status_event.query():
                _PENDING.remove(pg)
                pg._raise_for(int(pg._status_host.item()))

    def _raise_for(self, code):
        if code & 1:
            raise IndexError('edge_index contains node ids outside [0, n_nodes)')
        if code & 2:
            raise ValueError('edge_attr rows must be one-hot (the reference data loader emits '
                             'one_hot(edge_type, 3)); dense edge attributes are not supported')
        if code & 4:
            raise ValueError("the batch promises edge_layout == 'generate_edges' (per graph: two row-sorted runs, "
                             "node ids inside the graph's range) but its edge list is not laid out that way; "
                             "drop the tag (batch.edge_layout = None) to use the general sort")
        self._status_checked = True

    def check_status(self):
        """Host-side validation of the inputs (one tiny D2H copy; call where a sync is fine)."""
        if not self._status_checked:
            self._raise_for(int(self.t['status'].item()))


_PENDING = []


def runs_layout(graph, device=None):
    """(node_ptr, edge_ptr) device int32 tensors of a batch that carries the 'generate_edges' tag, else
    None. Cached on the batch object (one small host-to-device copy per batch)."""
    if getattr(graph, 'edge_layout', None) != 'generate_edges' or os.environ.get('PVS_PREPARE_RUNS') == '0':
        return None
    n_edges = int(graph.edge_index.size(1))
    cached = getattr(graph, '_runs_layout', None)
    if (cached is not None and cached[2] == (n_edges, graph.edge_index.data_ptr())
            and (device is None or cached[0].device == torch.device(device))):
        return cached[:2]
    counts, ptr = getattr(graph, 'graph_edge_counts', None), getattr(graph, 'ptr', None)
    if counts is None or ptr is None:
        return None
    # The device pass verifies the edge list AGAINST these two tables; the tables themselves are host data
    # and are checked here (free): an edge list filtered or extended after collation no longer matches its
    # stale per-graph counts, and the placement would then write outside the rows' slots. Such a batch
    # takes the general sort path.
    nodes = getattr(graph, 'graph_node_counts', None)
    if nodes is None:
        if ptr.is_cuda:          # only a device copy of the node table: not verifiable without a sync
            return None
        nodes = (ptr[1:] - ptr[:-1]).tolist()
    n_nodes = int(graph.x.size(0)) if torch.is_tensor(getattr(graph, 'x', None)) else sum(nodes)
    if (not counts or len(nodes) != len(counts) or sum(counts) != n_edges or min(counts) < 0
            or sum(nodes) != n_nodes or min(nodes) < 0):
        return None
    dev = graph.edge_index.device if device is None else device
    edge_ptr = torch.tensor([0] + list(counts), dtype=torch.int64).cumsum(0).to(dtype=torch.int32)
    node_ptr = torch.tensor([0] + list(nodes), dtype=torch.int64).cumsum(0).to(dtype=torch.int32)
    layout = (node_ptr.to(dev, non_blocking=True), edge_ptr.to(dev, non_blocking=True))
    # (host knowledge the device pass can use: with a bound on the graphs' sizes the by-column lists need no sort)
    layout[0].max_graph_nodes = int(max(nodes))
    graph._runs_layout = layout + ((n_edges, graph.edge_index.data_ptr()),)
    return layout


def prepare_graph(edge_index, edge_attr, n_nodes, need_backward=None, layout=None):
    """int64 COO `[2,E]` (+ int64 one-hot `[E,A]` or None) -> PreparedGraph on the same device.
    need_backward=False (default: torch.is_grad_enabled()) skips the by-column lists that only the
    backward reads. layout = runs_layout(batch): the batch's edge list has the reference loader's
    layout (two row-sorted runs per graph), prepared by merging instead of sorting (checked on the device)."""
    if need_backward is None:
        need_backward = torch.is_grad_enabled()
    _lib.require_hip(edge_index, edge_attr)
    lib = _lib.lib()
    if edge_index.dtype != torch.int64:
        edge_index = edge_index.long()
    edge_index = edge_index.contiguous()
    n_edges = int(edge_index.shape[1])
    n_attr = 0
    if edge_attr is not None:
        if edge_attr.dim() != 2 or edge_attr.shape[0] != n_edges:
            raise ValueError(f'edge_attr shape {tuple(edge_attr.shape)} does not match E={n_edges}')
        if edge_attr.dtype.is_floating_point:
            if not bool(((edge_attr == 0) | (edge_attr == 1)).all()):
                raise ValueError('edge_attr must be one-hot; dense edge attributes are not supported')
        edge_attr = edge_attr.long().contiguous()
        n_attr = int(edge_attr.shape[1])
    dev = edge_index.device
    e_alloc = max(n_edges, 1)
    i32 = dict(dtype=torch.int32, device=dev)
    t = {
        'rowptr': torch.empty(n_nodes + 1, **i32), 'row': torch.empty(e_alloc, **i32),
        'col': torch.empty(e_alloc, **i32), 'perm': torch.empty(e_alloc, **i32),
        'inv_deg': torch.empty(n_nodes, dtype=torch.float32, device=dev),
        'status': torch.empty(1, **i32),
    }
    if n_attr:
        t['etype'] = torch.empty(e_alloc, dtype=torch.uint8, device=dev)
    if need_backward:
        t['colptr'], t['cedge'] = torch.empty(n_nodes + 1, **i32), torch.empty(e_alloc, **i32)
    stream = _lib.stream(dev)
    outputs = (_lib.ptr(t['rowptr']), _lib.ptr(t['row']), _lib.ptr(t['col']), _lib.ptr(t.get('etype')),
               _lib.ptr(t['perm']), _lib.ptr(t.get('colptr')), _lib.ptr(t.get('cedge')), _lib.ptr(t['inv_deg']),
               _lib.ptr(t['status']))
    if layout is not None and n_edges > 0 and int(layout[0].numel()) == int(layout[1].numel()) >= 2:
        node_ptr, edge_ptr = layout
        n_graphs = int(node_ptr.numel()) - 1
        max_nodes = int(getattr(node_ptr, 'max_graph_nodes', 0)) if need_backward else 0
        ws_bytes = lib.pvs_graph_prepare_runs_workspace_bytes(n_nodes, n_edges, n_graphs, max_nodes)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        rc = lib.pvs_graph_prepare_runs(
            _lib.ptr(edge_index), _lib.ptr(edge_attr), n_attr, n_nodes, n_edges, n_graphs, _lib.ptr(node_ptr),
            _lib.ptr(edge_ptr), *outputs, max_nodes, _lib.ptr(ws), ws_bytes, stream)
        _lib.check(rc, 'pvs_graph_prepare_runs')
        t['_layout'] = layout      # (keeps the pointer tables alive until the kernels have run)
        # the batch's graphs are known: graph g's edges are the sorted positions [edge_ptr[g], edge_ptr[g + 1]) (graphs
        # hold consecutive node ranges and the list is sorted by row) - exactly PvsGraph.graph_eptr, so every caller
        # of this path gets tiles that end at graph boundaries, not only the models' forward
        if n_graphs >= 2:
            t['graph_eptr'] = edge_ptr
    else:
        ws_bytes = lib.pvs_graph_prepare_workspace_bytes(n_nodes, n_edges)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        rc = lib.pvs_graph_prepare(
            _lib.ptr(edge_index), _lib.ptr(edge_attr), n_attr, n_nodes, n_edges, *outputs, _lib.ptr(ws), ws_bytes,
            stream)
        _lib.check(rc, 'pvs_graph_prepare')
    pg = PreparedGraph(n_nodes, n_edges, n_attr, t)
    if 'graph_eptr' in t:
        pg.c.graph_eptr = _lib.ptr(t['graph_eptr'])
        pg.c.n_graphs = int(t['graph_eptr'].numel()) - 1
    return pg


_PREFETCH = {}
_PREFETCH_STREAM = {}


def _graph_key(edge_index, edge_attr, n_nodes):
    return (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape),
            None if edge_attr is None else (edge_attr.data_ptr(), edge_attr._version), n_nodes)


def prefetch_graph(edge_index, edge_attr, n_nodes, layout=None):
    """Start `prepare_graph` for an upcoming batch on a side stream (e.g. while the current batch
    is in its backward pass: the sorts are memory/latency-bound, the edge kernels ALU-bound). The
    next `prepared_for` call with the same tensors picks the result up and orders the consumer
    stream behind it. One outstanding prefetch per device."""
    dev = edge_index.device
    side = _PREFETCH_STREAM.get(dev)
    if side is None:
        side = _PREFETCH_STREAM[dev] = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))   # inputs may have been produced just now
    with torch.cuda.stream(side):
        pg = prepare_graph(edge_index, edge_attr, n_nodes, need_backward=True, layout=layout)
        done = torch.cuda.Event()
        done.record(side)
    _PREFETCH[dev] = (_graph_key(edge_index, edge_attr, n_nodes), pg, done, edge_index, edge_attr)
    return pg


def _take_prefetched(edge_index, edge_attr, n_nodes):
    hit = _PREFETCH.get(edge_index.device)
    if hit is None or hit[0] != _graph_key(edge_index, edge_attr, n_nodes):
        return None
    del _PREFETCH[edge_index.device]
    _, pg, done, _, _ = hit
    torch.cuda.current_stream(edge_index.device).wait_event(done)
    for t in pg.t.values():    # the side stream allocated these: tell the allocator who uses them now
        if torch.is_tensor(t):
            t.record_stream(torch.cuda.current_stream(edge_index.device))
    return pg


_CACHE = collections.OrderedDict()
_CACHE_SIZE = 4
CACHE_ENABLED = True   # bench.py turns this off: a training step prepares every batch afresh


def prepared_for(edge_index, edge_attr, n_nodes, layout=None):
    """Cached `prepare_graph`: the L layers of a forward are called with the same edge tensors."""
    pre = _take_prefetched(edge_index, edge_attr, n_nodes)
    if pre is not None:
        return pre
    if not CACHE_ENABLED:
        return prepare_graph(edge_index, edge_attr, n_nodes, layout=layout)
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape),
           None if edge_attr is None else (edge_attr.data_ptr(), edge_attr._version), n_nodes,
           torch.is_grad_enabled())      # a forward-only graph has no by-column lists
    hit = _CACHE.get(key)
    if hit is not None:
        _CACHE.move_to_end(key)
        return hit[0]
    pg = prepare_graph(edge_index, edge_attr, n_nodes, layout=layout)
    _CACHE[key] = (pg, edge_index, edge_attr)  # hold the inputs so data_ptr keys stay unique
    while len(_CACHE) > _CACHE_SIZE:
        _CACHE.popitem(last=False)
    return pg
