"""Training / validation harness that calls the path (caller, kept by surface).

Mirrors the constructor, loss, optimiser step and checkpoint format of
/root/reference/point_vs/models/point_neural_network_base.py:
  __init__        :49-123     get_loss   :362-370     backprop      :417-429
  train_model     :136-205    save       :501-517     load_weights  :528-565
  param_count / set_task :567-582
Progress bars, wandb and the predictions-file writer are outside the hot-path scope and reduced to
plain logging.
"""
import contextlib
import gc
import itertools
import math
import os
import re
import sys
import time
from abc import abstractmethod
from collections import OrderedDict
from pathlib import Path

import torch
import yaml
from torch import nn

from . import functional as PF
from .global_objects import DEVICE
from .optim import FusedClipAdam


def _rank_world():
    """(rank, world) of the default process group; (0, 1) for a single process."""
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1



@contextlib.contextmanager
def long_lived_heap_frozen():
    """`import torch` alone leaves ~170,000 container objects on the heap, and Python's first full (generation-2)
    collection walks all of them: 60-170 ms on the host, once, some tens of steps into a run (later ones are rare: they
    wait for a quarter as many NEW long-lived objects). Nothing for the reference's loop; a hundred steps' worth where a
    step takes ~1 ms (found in round 6: `PVS_BENCH_TRACE_STEPS=1 python bench.py --config real4A` prints the pause).
    gc.freeze() moves what exists now - modules, the model, the loader - out of the collector's sight for the duration of
    the loop; young objects are collected as before. PVS_GC_FREEZE=0: off."""
    global _FROZEN_DEPTH
    if os.environ.get('PVS_GC_FREEZE') == '0' or _FROZEN_DEPTH:      # (nested: validation inside a training run)
        yield
        return
    # (no gc.collect() first: a full collection is the very pause this avoids, and ScreeningSweep.run enters here once per
    # sweep inside callers' timed regions; cycles that are garbage right now stay until the loop ends)
    gc.freeze()
    _FROZEN_DEPTH = 1
    try:
        yield
    finally:
        _FROZEN_DEPTH = 0
        gc.unfreeze()


_FROZEN_DEPTH = 0

class _StepReplayer:
    """train_model(capture=True): whole training steps (graph preparation, forward, loss, backward, clip + Adam) as
    hipGraphs, one per batch that comes back with the same device tensors. A batch runs eagerly on its first visit
    (which also warms up whatever its shapes need), is captured and replayed on its second, replayed from then on; at
    most `max_graphs` batches are captured, the rest stay eager. Everything - eager steps too - runs on ONE side stream:
    autograd's AccumulateGrad nodes remember the stream of their first backward, and a capture on another stream than
    earlier eager steps faults in hipStreamEndCapture (ROCm 7.2 / torch 2.10; bench.py --graph 1 does the same).
    The optimiser runs in its capturable form for the duration (step counters on the device, as torch's capturable Adam
    keeps them; FusedClipAdam then forms the bias corrections in its kernel: optim.py) and is restored on close()."""

    def __init__(self, model, max_graphs=64):
        if not torch.cuda.is_available():
            raise RuntimeError('train_model(capture=True) needs a GPU')
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise NotImplementedError('train_model(capture=True) is single-process (the gradient exchange is not captured)')
        if model.scheduler is not None:
            raise NotImplementedError('train_model(capture=True) with a learning-rate scheduler (the rate is baked '
                                      'into the captured optimiser step)')
        if not isinstance(model.optimiser, torch.optim.Adam):
            raise NotImplementedError('train_model(capture=True) needs the Adam optimiser')
        self.model, self.max_graphs = model, max_graphs
        # one side stream per model, kept across calls (AccumulateGrad nodes remember the stream of their first backward)
        if getattr(model, '_capture_stream', None) is None:
            model._capture_stream = torch.cuda.Stream(DEVICE)
        self.stream = model._capture_stream
        self.seen, self.graphs = {}, {}
        self.stats = {'eager': 0, 'captured': 0, 'replayed': 0}
        model.last_capture_stats = self.stats
        self._was = []
        model._capturable_was = self._was       # save() writes the optimiser's OWN form, not the replayer's
        for group in model.optimiser.param_groups:
            self._was.append(group.get('capturable', False))
            group['capturable'] = True
            for p in group['params']:
                st = model.optimiser.state.get(p)
                if st and isinstance(st.get('step'), torch.Tensor) and not st['step'].is_cuda:
                    st['step'] = st['step'].to(device=p.device, dtype=torch.float32)
        model.optimiser._fast = None
        if hasattr(model.optimiser, 'reserve_capture_tables'):
            model.optimiser.reserve_capture_tables(max_graphs)

    def close(self):
        torch.cuda.current_stream(DEVICE).wait_stream(self.stream)
        self.graphs.clear()
        opt = self.model.optimiser
        for group, was in zip(opt.param_groups, self._was):
            group['capturable'] = was
            if not was:
                for p in group['params']:
                    st = opt.state.get(p)
                    if st and isinstance(st.get('step'), torch.Tensor) and st['step'].is_cuda:
                        st['step'] = st['step'].cpu()
        opt._fast = None
        self.model._capturable_was = None

    _batch_ids = itertools.count(1)

    @classmethod
    def _key(cls, graph, task):
        """A batch is 'the same batch again' when it is the same OBJECT with the same device tensors at the same
        versions: the object carries an id stamped on its first visit (a streaming loader that yields fresh batch
        objects whose tensors happen to reuse freed addresses is never taken for a second visit)."""
        try:
            bid = graph.__dict__.get('_pvs_batch_id')
            if bid is None:
                bid = graph.__dict__['_pvs_batch_id'] = next(cls._batch_ids)
        except AttributeError:
            return None
        key = [task, bid]
        for name in ('x', 'pos', 'edge_index', 'edge_attr', 'batch', 'y'):
            t = getattr(graph, name, None)
            if t is None:
                key.append(None)
                continue
            if not torch.is_tensor(t) or not t.is_cuda:
                return None                  # host tensors are copied to new device tensors every step: nothing to replay
            key.append((t.data_ptr(), t._version, tuple(t.shape)))
        return tuple(key)

    def _eager(self, graph):
        y_pred, y_true, _, _ = self.model.unpack_input_data_and_predict(graph)
        return self.model.backprop(y_true, y_pred, sync=False)

    @staticmethod
    def _make_capturable(graph):
        """What the forward would otherwise derive from the batch vector with data-dependent ops (a bincount, a
        maximum read on the host): done once, outside any capture, and left on the batch object."""
        batch = getattr(graph, 'batch', None)
        if batch is None:
            return
        if getattr(graph, 'num_graphs', None) is None:
            graph.num_graphs = int(batch.max()) + 1
        if getattr(graph, 'ptr', None) is None:
            counts = torch.bincount(batch, minlength=graph.num_graphs)
            graph.ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])

    def step(self, graph):
        from . import graph as pgraph
        key = self._key(graph, self.model.model_task)
        hit = self.graphs.get(key) if key is not None else None
        if hit is not None:
            hit[0].replay()
            self.stats['replayed'] += 1
            return hit[1].clone()
        visits = self.seen.get(key, 0) if key is not None else 0
        if key is None or visits == 0 or len(self.graphs) >= self.max_graphs:
            if key is not None:
                self.seen[key] = visits + 1
                self._make_capturable(graph)
            self.stats['eager'] += 1
            return self._eager(graph)
        # second visit: capture. The batch's CSR / CSC preparation is captured too (its tensors then live in the
        # graph's private pool: a cached PreparedGraph could be evicted and freed under the captured launches).
        cache_was, pgraph.CACHE_ENABLED = pgraph.CACHE_ENABLED, False
        try:
            self.stream.synchronize()
            hip_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(hip_graph, stream=self.stream):
                static_loss = self._eager(graph)
        finally:
            pgraph.CACHE_ENABLED = cache_was
        self.graphs[key] = (hip_graph, static_loss, graph)      # (holds the batch: its addresses stay unique)
        hip_graph.replay()
        self.stats['captured'] += 1
        return static_loss.clone()


class PointNeuralNetworkBase(nn.Module):
    """Base (abstract) class of the point-cloud networks."""

    def __init__(self, save_path, learning_rate, weight_decay=None, wandb_project=None,
                 wandb_run=None, silent=False, use_1cycle=False, warm_restarts=False,
                 only_save_best_models=False, optimiser='adam', regression_loss='mse',
                 **model_kwargs):
        super().__init__()
        self.set_task(model_kwargs.get('model_task', 'classification'))
        self.include_strain_info = False
        self.batch = 0
        self.p_epoch = 0
        self.a_epoch = 0
        self.save_path = Path(save_path).expanduser()
        self.only_save_best_models = only_save_best_models
        if not silent:
            self.save_path.mkdir(parents=True, exist_ok=True)
        self.predictions_file = Path(self.save_path, 'predictions.txt')
        self.lr = learning_rate
        self.weight_decay = weight_decay
        self.bce = nn.BCEWithLogitsLoss()
        self.regression_loss = nn.MSELoss() if regression_loss == 'mse' else nn.HuberLoss()
        self.wandb_project, self.wandb_run = wandb_project, wandb_run
        self.n_layers = model_kwargs.get('num_layers', 12)
        self.layers = self.build_net(**model_kwargs)
        if optimiser == 'adam':
            # torch.optim.Adam subclass: same rule and state_dict, one launch for clip + step on the GPU
            self.optimiser = FusedClipAdam(self.parameters(), lr=self.lr, weight_decay=weight_decay or 0)
        elif optimiser == 'sgd':
            self.optimiser = torch.optim.SGD(self.parameters(), lr=self.lr, momentum=0.9,
                                             weight_decay=weight_decay or 0, nesterov=True)
        else:
            raise NotImplementedError(f'{optimiser} not recognised optimiser.')
        assert not (use_1cycle and warm_restarts), '1cycle and warm restarts are mutually exclusive'
        self.use_1cycle, self.warm_restarts = use_1cycle, warm_restarts
        self.global_iter = 0
        self.val_iter = 0
        self.log_interval = 10
        self.scheduler = None
        self.test_metric = 0
        self.grad_sync = None   # optional callable run between backward and clip (data parallel)
        if not silent:
            with open(self.save_path / 'model_kwargs.yaml', 'w', encoding='utf-8') as f:
                yaml.dump(model_kwargs, f)
        self.to(DEVICE)

    @abstractmethod
    def build_net(self, **model_kwargs):
        """Construct self.layers (and the head)."""

    @abstractmethod
    def unpack_input_data_and_predict(self, input_data):
        """Unpack the graph into tensors and run the network."""

    def get_loss(self, y_true, y_pred):
        if self.model_task == 'classification':
            if (y_pred.is_cuda and y_pred.dtype == torch.float32      # nn.BCEWithLogitsLoss() as one launch each way
                    and os.environ.get('PVS_FUSED_HEAD') != '0'):
                return PF.bce_with_logits_mean(y_pred, y_true.to(device=y_pred.device, dtype=torch.float32))
            return self.bce(y_pred, y_true.to(y_pred.device))
        if self.model_task == 'regression':
            return self.regression_loss(y_pred, y_true.to(y_pred.device))
        y_pred[torch.where(y_true == -1)] = -1
        return 3 * self.regression_loss(y_pred, y_true.to(y_pred.device))

    def backprop(self, y_true, y_pred, sync=True):
        """One optimisation step (:417-429). sync=True is the reference's contract: returns the loss
        as a float and stops on NaN, at the price of a host round trip per step. sync=False returns
        the loss tensor and leaves the check to the caller (train_model checks every
        `log_interval` steps, so the GPU never waits for the host inside the loop)."""
        loss = self.get_loss(y_true, y_pred)
        self.optimiser.zero_grad()
        if loss.is_cuda and loss.dtype == torch.float32 and loss.dim() == 0:
            loss.backward(gradient=PF.unit_gradient(loss.device))     # (= loss.backward() without the root fill)
        else:
            loss.backward()
        if self.grad_sync is not None:
            self.grad_sync()
        if isinstance(self.optimiser, FusedClipAdam):
            self.optimiser.step(clip_value=1.0)
        else:
            torch.nn.utils.clip_grad_value_(self.parameters(), 1.0)
            self.optimiser.step()
        if not sync:
            return loss.detach()
        loss_ = float(loss.detach().cpu())
        if math.isnan(loss_):
            raise FloatingPointError('We have hit a NaN loss value.')
        return loss_

    @staticmethod
    def _drain_losses(pending, losses):
        """Device loss tensors -> floats (one transfer), NaN check as in backprop."""
        if pending:
            vals = torch.stack(pending).cpu().tolist()
            pending.clear()
            if any(math.isnan(v) for v in vals):
                raise FloatingPointError('We have hit a NaN loss value.')
            losses.extend(vals)

    def training_setup(self, data_loader, epochs, model_task=None):
        if self.use_1cycle:
            self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
                self.optimiser, max_lr=self.lr, steps_per_epoch=epochs * len(data_loader), epochs=1)
        elif self.warm_restarts:
            self.scheduler = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(
                self.optimiser, T_0=len(data_loader), T_mult=1, eta_min=0)
        if model_task is not None:
            self.set_task(model_task)
        init_epoch = self.a_epoch if 'regression' in self.model_task else self.p_epoch
        return init_epoch, time.time()

    def train_model(self, data_loader, epochs=1, epoch_end_validation_set=None, top1_on_end=False, capture=False):
        """Training loop (:136-205): per batch predict + backprop, per epoch checkpoint (+ optional
        validation, :470-490). No host synchronisation inside the loop: losses are drained every
        `log_interval` steps.
        capture=True (no reference counterpart, opt-in): batches that COME BACK - the same device tensors epoch after
        epoch, e.g. a list of pre-collated batches resident on the GPU - are replayed from a hipGraph of their whole
        step from their third visit on (`_StepReplayer`). For small graphs the eager step is bound by the host
        (~70 launches of a few microseconds: 16.8k -> 25.7k graphs/s at the reference's default shape,
        profiles/r05_real_shape.txt); for BASELINE-size batches it buys nothing. Single process, no learning-rate
        scheduler, Adam; anything else raises."""
        init_epoch, _ = self.training_setup(data_loader=data_loader, epochs=epochs)
        if capture:
            replayer = _StepReplayer(self)
            try:
                with torch.cuda.stream(replayer.stream), long_lived_heap_frozen():
                    return self._train_epochs(data_loader, init_epoch, epochs, epoch_end_validation_set, top1_on_end,
                                              replayer)
            finally:
                replayer.close()
        with long_lived_heap_frozen():
            return self._train_epochs(data_loader, init_epoch, epochs, epoch_end_validation_set, top1_on_end, None)

    def _train_epochs(self, data_loader, init_epoch, epochs, epoch_end_validation_set, top1_on_end, replayer):
        losses = []
        sampler = getattr(data_loader, 'sampler', None)
        for epoch in range(init_epoch, epochs):
            self.train()
            if hasattr(sampler, 'set_epoch'):      # per-rank seeded draw (pointvs_amd/data_loaders.py)
                sampler.set_epoch(epoch)
            pending = []
            for self.batch, graph in enumerate(data_loader):
                if replayer is not None:
                    pending.append(replayer.step(graph))
                else:
                    y_pred, y_true, _, _ = self.unpack_input_data_and_predict(graph)
                    if hasattr(self.grad_sync, 'set_weight'):     # uneven shards: weight by local graphs
                        self.grad_sync.set_weight(float(y_pred.numel()))
                    pending.append(self.backprop(y_true, y_pred, sync=False))
                if self.scheduler is not None:
                    self.scheduler.step()
                self.global_iter += 1
                if len(pending) >= self.log_interval:
                    self._drain_losses(pending, losses)
            self._drain_losses(pending, losses)
            self.eval()
            if 'regression' in self.model_task:
                self.a_epoch += 1
            else:
                self.p_epoch += 1
            if not self.only_save_best_models:
                self.save()
            done = self.a_epoch if 'regression' in self.model_task else self.p_epoch
            if epoch_end_validation_set is not None and done < epochs:     # (:481: not after the last epoch)
                fname = Path(self.predictions_file.parent, f'predictions_epoch_{done}.txt')
                best = self.val(epoch_end_validation_set, predictions_file=fname, top1_on_end=top1_on_end)
                if self.only_save_best_models and best:
                    self.save()
        return losses

    @torch.no_grad()
    def val(self, data_loader, predictions_file=None, top1_on_end=False, rich_ctx=None):
        """Inference loop (:208-360): writes `<save_path>/<pose|affinity>_<predictions file name>` in
        the reference's line format. The loop never waits for the host: scores leave the device
        through pinned buffers and a writer thread formats and appends them every `log_interval`
        batches, as the reference's write_predictions does (pointvs_amd/predictions.py).
        top1_on_end: the reference's model selection on the finished file (top-1 / Pearson); returns
        False only when `only_save_best_models` is set and this epoch is not the best so far."""
        from .predictions import PredictionsWriter, merge_rank_files, rank_part
        predictions_file = Path(predictions_file or self.predictions_file)
        predictions_file = (predictions_file.parent /
                            f'{self.model_task_for_fnames}_{predictions_file.name}').expanduser()
        rank, world = _rank_world()
        # one process per GPU: every rank scores its own contiguous share of the set (point_vs.py
        # make_loader) into its OWN part file; rank 0 joins the parts in rank order after a barrier
        part = predictions_file if world == 1 else rank_part(predictions_file, rank)
        self.eval()
        self.val_iter = 0
        with PredictionsWriter(part, self.model_task, flush_every=self.log_interval) as writer, long_lived_heap_frozen():
            for self.batch, graph in enumerate(data_loader):
                self.val_iter += 1
                y_pred, y_true, ligands, receptors = self.unpack_input_data_and_predict(graph)
                if self.model_task == 'classification':
                    y_pred = torch.sigmoid(y_pred)
                writer.submit(y_pred, y_true, receptors, ligands)
        if world > 1:
            torch.distributed.barrier()         # every part is complete and closed
            if rank == 0:
                merge_rank_files(predictions_file, world)
            torch.distributed.barrier()         # nobody reads the file before it is whole
        self.last_predictions_file = predictions_file
        if not top1_on_end:
            return True
        # model selection (:330-360): top-1 over receptors for pose models, Pearson's r (p < 0.05) for affinity
        # models, against the best value so far; every rank reads the same joined file and decides alike
        from .predictions import regression_pearson, top_n
        if self.model_task == 'classification':
            metric = top_n(predictions_file)
            best = metric > self.test_metric
        else:
            metric, p_value = regression_pearson(predictions_file)
            best = bool(p_value < 0.05 and metric > self.test_metric)
        if best:
            self.test_metric = float(metric)
        self.last_validation_metric = float(metric)
        return bool(best or not self.only_save_best_models)

    def save(self, save_path=None):
        """Checkpoint in the reference's format (:501-517). Data parallel: weights and optimiser state
        are identical on every rank, so rank 0 alone writes (ranks writing one path at once corrupt it)."""
        if _rank_world()[0] != 0:
            return
        epoch = self.a_epoch if 'regression' in self.model_task else self.p_epoch
        if save_path is None:
            save_path = (self.save_path / 'checkpoints' /
                         f'{self.model_task_for_fnames}_ckpt_epoch_{epoch}.pt')
        Path(save_path).parent.mkdir(parents=True, exist_ok=True)
        torch.save({
            'learning_rate': self.lr, 'weight_decay': self.weight_decay,
            'p_epoch': self.p_epoch, 'a_epoch': self.a_epoch,
            'model_state_dict': self.state_dict(),
            'optimiser_state_dict': self._optimiser_state_for_checkpoint()}, save_path)

    def _optimiser_state_for_checkpoint(self):
        """`optimiser.state_dict()` in the form the optimiser has OUTSIDE train_model(capture=True): the replayer
        flips every group to `capturable=True` with the step counters on the device for the duration of the call, and a
        checkpoint written at an epoch end in between would carry that (a resumed run would silently be capturable -
        another FusedClipAdam path, and torch's Adam asserts on CPU parameters). The groups get their own flags back
        and the counters go to the host, as the reference's checkpoints hold them (:501-517)."""
        sd = self.optimiser.state_dict()
        was = getattr(self, '_capturable_was', None)
        if not was:
            return sd
        groups = []
        for group, flag in zip(sd['param_groups'], was):
            group = dict(group)
            group['capturable'] = flag
            groups.append(group)
            if flag:
                continue
            for idx in group['params']:
                st = sd['state'].get(idx)
                if st is not None and torch.is_tensor(st.get('step')) and st['step'].is_cuda:
                    st = dict(st)
                    st['step'] = st['step'].detach().cpu()
                    sd['state'][idx] = st
        sd['param_groups'] = groups
        return sd

    @staticmethod
    def _transform_names(d):
        """Legacy key names of older reference checkpoints (:520-526)."""
        out = OrderedDict()
        for key, value in d.items():
            out[key.replace('edge_attention_mlp', 'att_mlp').replace(
                'node_attention_mlp', 'node_att_mlp')] = value
        return out

    def load_weights(self, checkpoint_file, silent=False):
        """Restore a checkpoint written by save() or by the reference (:528-565). Same-task
        checkpoints restore weights, optimiser state and epoch counters; a checkpoint of the OTHER
        task (per the model_kwargs.yaml two levels above it) only has its tensors copied by name,
        leaving optimiser and epochs alone, as the reference does. Legacy key names
        (`edge_attention_mlp`, `node_attention_mlp`, the 4-element `att_mlp` Sequential with its
        Linear at index 2) are tried only after a plain strict load fails."""
        checkpoint_file = Path(checkpoint_file).expanduser()
        if checkpoint_file.is_dir():
            found = sorted(checkpoint_file.glob('**/*.pt'), key=lambda p: p.stat().st_mtime)
            checkpoint_file = found[-1]
        checkpoint = torch.load(str(checkpoint_file), map_location=DEVICE)
        state = checkpoint['model_state_dict']
        kwargs_file = checkpoint_file.parents[1] / 'model_kwargs.yaml' if len(checkpoint_file.parents) > 1 else None
        ckpt_task = self.model_task
        if kwargs_file is not None and kwargs_file.is_file():
            ckpt_task = (yaml.safe_load(kwargs_file.read_text()) or {}).get('model_task', 'classification')
        if ckpt_task != self.model_task:
            own = self.state_dict()
            for name, value in state.items():
                own[name].copy_(value)
            return
        try:
            self.load_state_dict(state)
        except RuntimeError:
            renamed = self._transform_names(state)
            # the older 4-element att_mlp Sequential: Linear at index 2 (anchored: not node_att_mlp)
            renamed = OrderedDict((re.sub(r'(^|\.)att_mlp\.2\.', r'\1att_mlp.0.', k), v) for k, v in renamed.items())
            self.load_state_dict(renamed)
        if 'optimiser_state_dict' in checkpoint:
            try:
                self.optimiser.load_state_dict(checkpoint['optimiser_state_dict'])
            except ValueError as exc:     # e.g. a checkpoint of a model with a different parameter list
                print(f'pointvs_amd: optimiser state of {checkpoint_file} not restored ({exc})', file=sys.stderr)
        self.p_epoch = checkpoint.get('p_epoch', checkpoint.get('epoch', 0))
        self.a_epoch = checkpoint.get('a_epoch', 0)

    @property
    def param_count(self):
        return sum(torch.numel(t) for t in self.parameters() if t.requires_grad)

    def set_task(self, task):
        if task not in ('classification', 'regression', 'multi_regression'):
            raise ValueError('Argument for set_task must be one of classification, regression or '
                             'multi_regression')
        changed = getattr(self, 'model_task', task) != task
        self.model_task = task
        sync = getattr(self, 'grad_sync', None)
        if changed and hasattr(sync, 'reset'):
            # another head receives gradients from now on (MultitaskSatorrasEGNN: pose / affinity): the
            # exchange must learn its bucket layout again, identically on every rank
            sync.reset()
        if 'regression' in task:
            self.model_task_for_fnames = 'affinity'
            self.model_task_string = 'Mean squared error'
        else:
            self.model_task_for_fnames = 'pose'
            self.model_task_string = 'Binary crossentropy'
