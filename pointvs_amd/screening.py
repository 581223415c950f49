"""Virtual-screening forward (BASELINE config 5): many rigid poses of one
ligand against one receptor. Reference loop: `val` / inference.py
(/root/reference/point_vs/models/point_neural_network_base.py:208-360, point_vs/inference.py:35-146).

What is reused across poses: the receptor-receptor messages of the FIRST EGNN layer depend only on
receptor features and coordinates, so their per-node sums are computed once
(`pvs_egnn_layer_edge_sums` on the receptor-only graph); per batch the first layer only runs over
the edges that touch a ligand atom (`pvs_egnn_layer_fwd_partial`), the other layers over the full
graph. Graphs are built on the GPU from the coordinates (radius_graph.py). Scores equal the plain
`model(batch)` forward up to fp32 summation order.
"""
import ctypes as C

import torch

from . import _lib
from . import functional as PF
from .radius_graph import PoseBatcher, radius_graph


def _stream(dev):
    return _lib.stream(dev)


class ReceptorScreen:
    """scores = ReceptorScreen(model, rec_pos, feats, n_lig, batch_size, edge_radius)(lig_poses)

    feats [n_lig + n_rec, F]: ligand rows first, last column = bp (preprocessing.make_bit_vector);
    lig_poses [batch_size, n_lig, 3] on the device. Returns the model's raw outputs [batch_size, ...]."""

    def __init__(self, model, rec_pos, feats, n_lig, batch_size, edge_radius, intra_radius=None):
        layers = list(model.layers)
        self.model, self.embed, self.egnn = model, layers[0], layers[1:]
        first = self.egnn[0] if self.egnn else None
        self.reuse = (first is not None and first.hidden_nf in (32, 64) and not first.softmax_attention
                      and not first.edge_residual)
        dev = rec_pos.device
        self.batcher = PoseBatcher(rec_pos, feats, n_lig, batch_size, edge_radius, intra_radius)
        self.n_lig, self.b = n_lig, batch_size
        self.r_inter = edge_radius
        self.r_intra = edge_radius if intra_radius is None else intra_radius
        self._lig_buf, self._pending, self._l1_ws = None, None, None
        self._fast = None
        self._graph_ptr = self.batcher.batch.ptr.to(device=dev, dtype=torch.int32)
        # the specialised pose-batch builder (pvs_screen_graph_build) leaves the edge counts on the
        # device; layers that return edge messages (edge_residual) need them on the host
        self.fast_graph = self.reuse and n_lig <= 64 and not any(l.edge_residual for l in self.egnn)
        self._rec_pos, self._feats = rec_pos, feats
        if self.reuse:
            self._cache_receptor_sums()

    def _weights_fingerprint(self):
        """Version counters of everything the cached receptor-receptor sums were computed from (the
        embedding and the first layer): an optimiser step or load_weights() bumps them."""
        mods = [self.embed.m, self.egnn[0]]
        return tuple((p.data_ptr(), p._version) for m in mods for p in m.parameters())

    def _cache_receptor_sums(self):
        """Per-node sums of the first layer's receptor-receptor messages (pose independent)."""
        lib = _lib.lib()
        rec_pos, feats, first = self._rec_pos, self._feats, self.egnn[0]
        n_lig, batch_size = self.n_lig, self.b
        dev = rec_pos.device
        n_rec = rec_pos.shape[0]
        self._fingerprint = self._weights_fingerprint()
        with torch.no_grad():
            feats_rec = feats[n_lig:].to(dev).float()
            h_rec = self.embed.embed(feats_rec, rec_pos)
            pg = radius_graph(rec_pos, torch.ones(n_rec, dtype=torch.uint8, device=dev), None,
                              self.r_inter, self.r_intra, need_backward=False)
            desc = _lib.PvsLayerDesc(*first._desc())
            params = [None if p is None else p.detach().float().contiguous() for p in first._params()]
            pstruct = _lib.PvsLayerParams(*[_lib.ptr(p) for p in params])
            magg = torch.empty((n_rec, first.hidden_nf), dtype=torch.float32, device=dev)
            xsum = torch.empty((n_rec, 3), dtype=torch.float32, device=dev)
            ws_bytes = lib.pvs_egnn_layer_workspace_bytes(C.byref(desc), n_rec, pg.n_edges, 2)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.pvs_egnn_layer_edge_sums(
                C.byref(desc), C.byref(pg.c), C.byref(pstruct), _lib.ptr(h_rec.contiguous()),
                _lib.ptr(rec_pos.contiguous()), _lib.ptr(magg), _lib.ptr(xsum), _lib.ptr(ws), ws_bytes,
                _stream(dev)), 'pvs_egnn_layer_edge_sums')
            deg = (pg.t['rowptr'][1:] - pg.t['rowptr'][:-1]).float()
            self._rr = pg          # receptor-receptor template (rowptr / col, receptor-local ids)
            n = n_lig + n_rec
            # batch layout: [ligand rows (no base) | receptor rows] per pose
            self.base_magg = torch.zeros((batch_size, n, first.hidden_nf), dtype=torch.float32, device=dev)
            self.base_xsum = torch.zeros((batch_size, n, 3), dtype=torch.float32, device=dev)
            self.base_deg = torch.zeros((batch_size, n), dtype=torch.float32, device=dev)
            self.base_magg[:, n_lig:] = magg
            self.base_xsum[:, n_lig:] = xsum
            self.base_deg[:, n_lig:] = deg
            torch.cuda.current_stream(dev).synchronize()    # ws / pg go out of scope

    def _ligand_graph(self, batch):
        """CSR of the full graph's ligand-touching edges, filtered on the device (no host round trip);
        the edge count stays on the device (PvsGraph.n_edges_dev)."""
        lib = _lib.lib()
        full = batch.prepared
        dev = batch.pos.device
        n = full.n_nodes
        if self._lig_buf is None:
            # one synchronous probe sizes the buffers; later batches are checked asynchronously
            probe = radius_graph(batch.pos, batch.x[:, -1], batch.ptr, self.r_inter, self.r_intra,
                                 max_graph_nodes=self.batcher.n, need_backward=False, ligand_pairs_only=True)
            cap = min(4 * self.n_lig * self.batcher.n * self.b, 2 * probe.n_edges + 4096)
            i32 = dict(dtype=torch.int32, device=dev)
            self._lig_buf = dict(
                cap=cap, rowptr=torch.empty(n + 1, **i32), row=torch.empty(cap, **i32), col=torch.empty(cap, **i32),
                etype=torch.empty(cap, dtype=torch.uint8, device=dev), status=torch.zeros(1, **i32),
                ones=torch.ones(n, dtype=torch.float32, device=dev),
                ws=torch.empty(lib.pvs_graph_filter_workspace_bytes(n), dtype=torch.uint8, device=dev),
                bp=batch.x[:, -1].to(torch.uint8).contiguous(),
                host=torch.zeros(1, dtype=torch.int32).pin_memory())
        b = self._lig_buf
        self.check()      # the previous batch's overflow flag has landed by now
        _lib.check(lib.pvs_graph_filter_ligand_edges(
            C.byref(full.c), _lib.ptr(b['bp']), b['cap'], _lib.ptr(b['rowptr']), _lib.ptr(b['row']),
            _lib.ptr(b['col']), _lib.ptr(b['etype']), _lib.ptr(b['status']), _lib.ptr(b['ws']), b['ws'].numel(),
            _stream(dev)), 'pvs_graph_filter_ligand_edges')
        b['host'].copy_(b['status'], non_blocking=True)
        self._pending = torch.cuda.Event()
        self._pending.record(torch.cuda.current_stream(dev))
        g = _lib.PvsGraph()
        g.n_nodes, g.n_edges = n, b['cap']
        g.rowptr, g.row, g.col, g.etype = (_lib.ptr(b[k]) for k in ('rowptr', 'row', 'col', 'etype'))
        g.inv_deg = _lib.ptr(b['ones'])
        g.n_edges_dev = b['rowptr'][n:].data_ptr()
        return g

    def check(self):
        """Raises if the ligand-edge buffers of an earlier batch were too small (checked one batch
        late so that the loop never waits for the device; call once more after the last batch)."""
        if self._pending is not None:
            self._pending.synchronize()
            self._pending = None
            if int(self._lig_buf['host'].item()) & 4:
                raise RuntimeError('ReceptorScreen: ligand-edge buffer overflow (more ligand contacts than '
                                   'twice the first batch); rebuild the screen with a larger probe')

    def _first_layer(self, batch, h, x, g=None):
        lib = _lib.lib()
        first = self.egnn[0]
        dev = h.device
        if g is None:
            g = self._ligand_graph(batch)
        desc = _lib.PvsLayerDesc(*first._desc())
        params = [None if p is None else p.detach().float().contiguous() for p in first._params()]
        pstruct = _lib.PvsLayerParams(*[_lib.ptr(p) for p in params])
        n = h.shape[0]
        h_out, x_out = torch.empty_like(h), torch.empty_like(x)
        natt = torch.empty(n, dtype=torch.float32, device=dev) if first.node_attention else None
        if self._l1_ws is None:
            self._l1_ws = (
                torch.empty(lib.pvs_egnn_layer_saved_floats(C.byref(desc), n, g.n_edges), dtype=torch.float32,
                            device=dev),
                torch.empty(lib.pvs_egnn_layer_workspace_bytes(C.byref(desc), n, g.n_edges, 2), dtype=torch.uint8,
                            device=dev))
        saved, ws = self._l1_ws
        _lib.check(lib.pvs_egnn_layer_fwd_partial(
            C.byref(desc), C.byref(g), C.byref(pstruct), _lib.ptr(h), _lib.ptr(x),
            _lib.ptr(self.base_magg), _lib.ptr(self.base_xsum), _lib.ptr(self.base_deg), _lib.ptr(h_out),
            _lib.ptr(x_out), _lib.ptr(natt), _lib.ptr(saved), _lib.ptr(ws), ws.numel(), _stream(dev)),
            'pvs_egnn_layer_fwd_partial')
        return h_out, x_out

    def _build_fast(self, lig_poses):
        """Full graph + ligand-touching subgraph of the pose batch from the receptor template
        (pvs_screen_graph_build): no receptor-receptor distance tests, no host round trip."""
        from .graph import PreparedGraph
        lib = _lib.lib()
        dev = lig_poses.device
        n, b_, n_rec = self.batcher.n, self.b, self.batcher.n - self.n_lig
        big_n = n * b_
        if self._fast is None:
            batch = self.batcher.load(lig_poses)       # one synchronous probe sizes the buffers
            probe = radius_graph(batch.pos, batch.x[:, -1], batch.ptr, self.r_inter, self.r_intra,
                                 max_graph_nodes=n, need_backward=False, ligand_pairs_only=True)
            cap_l = min(4 * self.n_lig * n * b_, 2 * probe.n_edges + 4096)
            cap = b_ * self._rr.n_edges + cap_l
            i32 = dict(dtype=torch.int32, device=dev)
            f = dict(cap=cap, cap_l=cap_l, status=torch.zeros(1, **i32),
                     host=torch.zeros(1, dtype=torch.int32).pin_memory(),
                     state=torch.empty(lib.pvs_screen_graph_state_bytes(b_, self.n_lig, n_rec), dtype=torch.uint8,
                                       device=dev))
            for tag, c in (('', cap), ('_l', cap_l)):
                f['rowptr' + tag] = torch.empty(big_n + 1, **i32)
                f['row' + tag] = torch.empty(c, **i32)
                f['col' + tag] = torch.empty(c, **i32)
                f['etype' + tag] = torch.empty(c, dtype=torch.uint8, device=dev)
            f['inv_deg'] = torch.empty(big_n, dtype=torch.float32, device=dev)
            f['ones'] = torch.ones(big_n, dtype=torch.float32, device=dev)
            pg = PreparedGraph(big_n, cap, 3, dict(rowptr=f['rowptr'], row=f['row'], col=f['col'],
                                                    etype=f['etype'], inv_deg=f['inv_deg'], status=f['status']))
            pg._status_checked = True
            pg.c.n_edges_dev = f['rowptr'][big_n:].data_ptr()
            gl = _lib.PvsGraph()
            gl.n_nodes, gl.n_edges = big_n, cap_l
            gl.rowptr, gl.row, gl.col, gl.etype = (_lib.ptr(f[k + '_l']) for k in ('rowptr', 'row', 'col', 'etype'))
            gl.inv_deg = _lib.ptr(f['ones'])
            gl.n_edges_dev = f['rowptr_l'][big_n:].data_ptr()
            f['pg'], f['gl'] = pg, gl
            self._fast = f
        f = self._fast
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.check()
        self.batcher._pos[:, :self.n_lig] = lig_poses
        _lib.check(lib.pvs_screen_graph_build(
            _lib.ptr(lig_poses.contiguous()), _lib.ptr(self.batcher._pos[0, self.n_lig:].contiguous()),
            _lib.ptr(self._rr.t['rowptr']), _lib.ptr(self._rr.t['col']), b_, self.n_lig, n_rec,
            float(self.r_inter), float(self.r_intra), f['cap'], f['cap_l'],
            _lib.ptr(f['rowptr']), _lib.ptr(f['row']), _lib.ptr(f['col']), _lib.ptr(f['etype']),
            _lib.ptr(f['inv_deg']), _lib.ptr(f['rowptr_l']), _lib.ptr(f['row_l']), _lib.ptr(f['col_l']),
            _lib.ptr(f['etype_l']), _lib.ptr(f['status']), _lib.ptr(f['state']), f['state'].numel(),
            _stream(dev)), 'pvs_screen_graph_build')
        self._lig_buf = dict(host=f['host'])       # check() reads the overflow flag from here
        if not capturing:
            self._poll_status()
        return f['pg'], f['gl']

    def _poll_status(self):
        f = self._fast
        f['host'].copy_(f['status'], non_blocking=True)
        self._pending = torch.cuda.Event()
        self._pending.record(torch.cuda.current_stream(f['status'].device))

    def capture(self, example_poses):
        """Captures one whole screening step (graph build + layer stack + head) in a hipGraph; after
        this, `replay(lig_poses)` copies the poses into the captured input buffer and launches the
        graph (BASELINE config 5: "hipGraph-captured layer stack"). Needs the device-side edge counts
        (self.fast_graph)."""
        if not self.fast_graph:
            raise RuntimeError('capture needs the pose-batch builder (<= 64 ligand atoms, no edge_residual)')
        dev = example_poses.device
        self._static_in = example_poses.clone()
        stream = torch.cuda.Stream(dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            for _ in range(2):
                self(self._static_in)          # warm-up: probe, buffers, lazy allocations
            self.check()
            torch.cuda.synchronize(dev)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, stream=stream):
                self._static_out = self(self._static_in)
        torch.cuda.current_stream(dev).wait_stream(stream)
        return self

    def stale(self):
        """True when the weights the cached receptor-receptor sums came from have changed since
        (load_weights(), an optimiser step): a captured step has those sums baked in."""
        return self.reuse and self._fingerprint != self._weights_fingerprint()

    def replay(self, lig_poses):
        if self.stale():           # host-only comparison of version counters
            raise RuntimeError('ReceptorScreen: the model\'s weights changed after capture(); build a new screen')
        self.check()
        self._static_in.copy_(lig_poses)
        self._graph.replay()
        self._poll_status()
        return self._static_out

    @torch.no_grad()
    def __call__(self, lig_poses):
        model = self.model
        if not self.reuse:
            return model(self.batcher.load(lig_poses))
        if self._fingerprint != self._weights_fingerprint():
            # the weights changed under the screen (load_weights(), a training step): the cached sums
            # are stale - recompute them once (never inside a captured graph, which bakes them in)
            if torch.cuda.is_current_stream_capturing() or getattr(self, '_graph', None) is not None:
                raise RuntimeError('ReceptorScreen: the model\'s weights changed after capture(); build a new screen')
            self._cache_receptor_sums()
        if self.fast_graph:
            pg_full, g_lig = self._build_fast(lig_poses)
            batch = self.batcher.batch
        else:
            batch = self.batcher.load(lig_poses)
            pg_full, g_lig = batch.prepared, None
        h = self.embed.embed(batch.x.float(), batch.pos).contiguous()
        x = batch.pos.contiguous()
        h, x = self._first_layer(batch, h, x, g_lig)
        m_sorted = None
        for layer in self.egnn[1:]:
            h, x, m_sorted = layer.forward_prepared(pg_full, h, x, m_sorted, need_m=layer.edge_residual,
                                                    need_coords=layer is not self.egnn[-1])
        if model.feats_linear_layers is None:
            return h
        return model._pool_and_head(model.feats_linear_layers, h, self._graph_ptr, self.b)


class ScreeningSweep:
    """Virtual-screening sweep (BASELINE config 5; the reference's `val` / inference.py loop,
    point_neural_network_base.py:208-360, inference.py:77-146): many ligands, each with many rigid
    poses, against ONE receptor, forward only.

    Poses are streamed in fixed-size batches through a `ReceptorScreen` per SIZE BUCKET (= number of
    ligand atoms: a captured hipGraph has fixed shapes, so every distinct ligand size gets its own
    captured step, built on first use and replayed for every later batch of that size; the ligand's
    features are written into the bucket's static input buffers). The last batch of a ligand is
    padded with copies of its last pose; the padding's scores are dropped. Scores leave the device
    through `PredictionsWriter` (pinned buffers + a writer thread, reference line format
    `'{score:.3f} | {receptor} {pose name}'`, :318-325), so the loop never waits for the host.

        sweep = ScreeningSweep(model, rec_pos, rec_feats, edge_radius=10.0, batch_size=32)
        scores = sweep.run([(name, lig_feats [n_lig,F], poses [P,n_lig,3]), ...], 'predictions.txt')
    """

    def __init__(self, model, rec_pos, rec_feats, edge_radius, batch_size=32, intra_radius=None, capture=True,
                 receptor_name='receptor'):
        self.model, self.rec_pos, self.rec_feats = model, rec_pos, rec_feats
        self.edge_radius, self.intra_radius, self.b = edge_radius, intra_radius, int(batch_size)
        self.capture, self.receptor_name = capture, receptor_name
        self.buckets = {}          # n_lig -> ReceptorScreen (captured when possible)
        self.batches_run = 0

    def _bucket(self, n_lig, lig_feats, example_poses):
        screen = self.buckets.get(n_lig)
        if screen is not None and screen.stale():
            # the model was trained on or reloaded since this bucket was captured: its cached receptor
            # sums (baked into the captured step) are out of date - build and capture it again
            screen = None
        if screen is None:
            feats = torch.cat([lig_feats.to(self.rec_feats.device), self.rec_feats], 0)
            screen = ReceptorScreen(self.model, self.rec_pos, feats, n_lig, self.b, self.edge_radius, self.intra_radius)
            screen._captured = False
            if self.capture and screen.fast_graph:
                screen.capture(example_poses)
                screen._captured = True
            self.buckets[n_lig] = screen
        return screen

    @staticmethod
    def _set_ligand_feats(screen, lig_feats):
        """The bucket's static node-feature buffer: ligand rows of every pose slot <- this ligand."""
        x = screen.batcher.batch.x
        x.view(screen.b, screen.batcher.n, -1)[:, :screen.n_lig] = lig_feats.to(x.device, x.dtype)

    @torch.no_grad()
    def run(self, ligands, predictions_file=None, sigmoid=None):
        """ligands: iterable of (name, lig_feats [n_lig,F], poses [P,n_lig,3] on the device).
        Returns {name: scores [P, ...] on the device} (raw model outputs; sigmoid-ed like `val` does for
        classification models when sigmoid is None/True). predictions_file: optional path."""
        from .predictions import PredictionsWriter
        if sigmoid is None:
            sigmoid = getattr(self.model, 'model_task', 'classification') == 'classification'
        writer = PredictionsWriter(predictions_file, 'regression', flush_every=10) if predictions_file else None
        out = {}
        from .point_neural_network_base import long_lived_heap_frozen
        frozen = long_lived_heap_frozen()       # (no full-heap collection pause inside the sweep: see its docstring)
        frozen.__enter__()
        try:
            for name, lig_feats, poses in ligands:
                n_poses, n_lig = int(poses.shape[0]), int(poses.shape[1])
                if n_poses == 0:
                    continue
                pad = (-n_poses) % self.b
                if pad:
                    poses = torch.cat([poses, poses[-1:].expand(pad, -1, -1)], 0)
                screen = self._bucket(n_lig, lig_feats, poses[:self.b].contiguous())
                self._set_ligand_feats(screen, lig_feats)
                scores = []
                for k in range(0, poses.shape[0], self.b):
                    chunk = poses[k:k + self.b]
                    y = screen.replay(chunk).clone() if screen._captured else screen(chunk.contiguous())
                    y = y.reshape(self.b, -1)
                    if sigmoid:
                        y = torch.sigmoid(y)
                    keep = min(self.b, n_poses - k)
                    scores.append(y[:keep])
                    if writer is not None:
                        writer.submit(y[:keep, 0], None, [self.receptor_name] * keep,
                                      [f'{name}_pose{k + i}' for i in range(keep)])
                    self.batches_run += 1
                out[name] = torch.cat(scores, 0)
                screen.check()
        finally:
            frozen.__exit__(None, None, None)
            if writer is not None:
                writer.close()
        return out
