"""The training harness against TRAJECTORIES of the reference's own `train_model`
(tests/golden/train_*.npz, written by tests/golden/make_golden_training.py from /root/reference): twelve optimiser
steps on a fixed list of batches under the three learning-rate schedules the reference offers, the multitask
sequence pose -> set_task('regression') -> affinity of point_vs.py:258-270, and BASELINE config 3's flag set at 64
channels (the H = 64 kernels in a training run).

Checked per step: the loss `backprop()` computes (1e-4 relative: twelve Adam steps amplify fp32 summation-order noise,
a single step is held to 1e-5 in tests/test_gpu_parity.py) and the learning rate the step ran at (the schedulers are
torch's own, so the sequence must be the reference's to the last bit of a float64); at the end: epoch counters,
global_iter, the checkpoint files and their dict keys, the optimiser's per-parameter step counts, and the weights.
Reference: point_neural_network_base.py:136-205, 372-388, 417-429, 470-517."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / 'golden'


def _load(name):
    z = np.load(GOLDEN / f'train_{name}.npz')
    return z, json.loads(str(z['meta']))


def _loaders(z, meta):
    from pointvs_amd.graph import Batch
    out = []
    for pi, (_, n_batches, _) in enumerate(meta['phases']):
        loader = []
        for bi in range(n_batches):
            pre = f'in/p{pi}b{bi}/'
            batch = torch.from_numpy(z[pre + 'batch'].astype(np.int64))
            n_graphs = int(batch.max()) + 1
            loader.append(Batch(
                x=torch.from_numpy(z[pre + 'x']), pos=torch.from_numpy(z[pre + 'pos']),
                edge_index=torch.from_numpy(z[pre + 'edge_index'].astype(np.int64)),
                edge_attr=torch.nn.functional.one_hot(torch.from_numpy(z[pre + 'edge_type'].astype(np.int64)), 3),
                batch=batch, y=torch.from_numpy(z[pre + 'y']), lig_fname=['l'] * n_graphs, rec_fname=['r'] * n_graphs,
                num_graphs=n_graphs))
        out.append(loader)
    return out


@pytest.mark.parametrize('name', ['default', 'one_cycle', 'warm_restarts', 'multitask', 'k64_attention'])
def test_training_trajectory_matches_the_reference(name, tmp_path):
    from pointvs_amd.egnn_multitask import MultitaskSatorrasEGNN
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    z, meta = _load(name)
    cls = SartorrasEGNN if meta['class'] == 'SartorrasEGNN' else MultitaskSatorrasEGNN
    torch.manual_seed(meta['seed'])
    np.random.seed(meta['seed'])
    model = cls(tmp_path, meta['lr'], meta['wd'], None, None, silent=True, **meta['ctor'], **meta['kwargs'])
    # the same seed gives the reference's initial weights bit for bit
    for k, v in model.state_dict().items():
        assert np.array_equal(v.detach().cpu().numpy(), z[f'sd0/{k}']), k

    lrs = []
    real_backprop = model.backprop

    def backprop(y_true, y_pred, sync=True):
        lrs.append(float(model.optimiser.param_groups[0]['lr']))
        return real_backprop(y_true, y_pred, sync=sync)
    model.backprop = backprop
    losses = []
    for (task, _, epochs), loader in zip(meta['phases'], _loaders(z, meta)):
        model.set_task(task)
        losses += [float(v) for v in model.train_model(loader, epochs=epochs)]

    ref_loss, ref_lr = z['loss'], z['lr']
    assert len(losses) == len(ref_loss) == len(lrs)
    assert np.array_equal(np.asarray(lrs, dtype=np.float64), ref_lr), (lrs, ref_lr.tolist())
    rel = np.abs(np.asarray(losses) - ref_loss) / np.abs(ref_loss)
    assert rel.max() < 1e-4, (rel.tolist(), losses, ref_loss.tolist())
    assert (model.p_epoch, model.a_epoch, model.global_iter) == (meta['p_epoch'], meta['a_epoch'], meta['global_iter'])
    ckpts = sorted(str(p.relative_to(tmp_path)) for p in tmp_path.rglob('*.pt'))
    assert ckpts == meta['checkpoints']
    ck = torch.load(tmp_path / ckpts[-1], map_location='cpu', weights_only=False)
    assert sorted(ck.keys()) == meta['checkpoint_keys']
    opt_steps = sorted({int(s['step']) for s in ck['optimiser_state_dict']['state'].values()})
    assert opt_steps == meta['optimiser_steps_in_last_checkpoint']
    # the weights after twelve steps: Adam moves a parameter by ~lr per step whatever its gradient's size, so an entry
    # whose gradient is rounding noise can end anywhere within steps * lr; everything else tracks the reference
    worst = 0.0
    for k, v in model.state_dict().items():
        got, ref = v.detach().cpu().numpy().astype(np.float64), z[f'sd1/{k}'].astype(np.float64)
        if not np.issubdtype(z[f'sd1/{k}'].dtype, np.floating):
            continue
        d = np.abs(got - ref)
        assert d.max() <= len(losses) * 2e-3 * 1.001, k
        worst = max(worst, float(np.quantile(d, 0.99)) if d.size > 20 else float(d.max()))
        moved = np.abs(ref - z[f'sd0/{k}'].astype(np.float64))
        assert np.median(d) <= 1e-3 * max(float(np.median(moved)), 1e-6) + 1e-6, (k, float(np.median(d)), float(np.median(moved)))
    print(f'{name}: worst loss rel {rel.max():.2e}, 99th percentile weight distance {worst:.2e}')


from tests._golden import needs_caching_allocator  # noqa: E402


@needs_caching_allocator
@pytest.mark.parametrize('name', ['default', 'multitask'])
def test_captured_training_steps_follow_the_reference_trajectory(name, tmp_path):
    """train_model(capture=True) (round 5, opt-in, no reference counterpart): batches that come back with the same
    device tensors are replayed from a hipGraph of their whole step. On the reference's trajectories (4 batches x 3
    epochs per phase) every batch runs eagerly in its first epoch, is captured in its second and replayed in its third;
    losses, learning rates, counters, checkpoints and optimiser step counts must be the eager run's = the reference's,
    and the optimiser must be back in its fused, non-capturable form afterwards."""
    from pointvs_amd.egnn_multitask import MultitaskSatorrasEGNN
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.optim import FusedClipAdam
    z, meta = _load(name)
    cls = SartorrasEGNN if meta['class'] == 'SartorrasEGNN' else MultitaskSatorrasEGNN
    torch.manual_seed(meta['seed'])
    np.random.seed(meta['seed'])
    model = cls(tmp_path, meta['lr'], meta['wd'], None, None, silent=True, **meta['ctor'], **meta['kwargs'])
    losses, stats = [], []
    for (task, n_batches, epochs), loader in zip(meta['phases'], _loaders(z, meta)):
        model.set_task(task)
        loader = [b.to('cuda') for b in loader]              # resident batches: the same tensors every epoch
        losses += [float(v) for v in model.train_model(loader, epochs=epochs, capture=True)]
        stats.append((dict(model.last_capture_stats), n_batches, epochs))
    for st, n_batches, epochs in stats:
        assert st['eager'] == n_batches and st['captured'] == (n_batches if epochs > 1 else 0), st
        assert st['replayed'] == n_batches * max(epochs - 2, 0), st
    ref_loss = z['loss']
    assert len(losses) == len(ref_loss)
    rel = np.abs(np.asarray(losses) - ref_loss) / np.abs(ref_loss)
    assert rel.max() < 1e-4, (rel.tolist(), losses, ref_loss.tolist())
    assert (model.p_epoch, model.a_epoch, model.global_iter) == (meta['p_epoch'], meta['a_epoch'], meta['global_iter'])
    ckpts = sorted(str(p.relative_to(tmp_path)) for p in tmp_path.rglob('*.pt'))
    assert ckpts == meta['checkpoints']
    ck = torch.load(tmp_path / ckpts[-1], map_location='cpu', weights_only=False)
    opt_steps = sorted({int(s['step']) for s in ck['optimiser_state_dict']['state'].values()})
    assert opt_steps == meta['optimiser_steps_in_last_checkpoint']
    for k, v in model.state_dict().items():
        ref = z[f'sd1/{k}']
        if np.issubdtype(ref.dtype, np.floating):
            assert np.abs(v.detach().cpu().numpy().astype(np.float64) - ref).max() <= len(losses) * 2e-3 * 1.001, k
    # back to the fused optimiser: host-side step counters, not capturable, and a further eager step works
    assert isinstance(model.optimiser, FusedClipAdam)
    assert not any(g.get('capturable') for g in model.optimiser.param_groups)
    assert all(not s['step'].is_cuda for s in model.optimiser.state.values())
    more = model.train_model([_loaders(z, meta)[-1][0]], epochs=model.a_epoch + model.p_epoch + 1)
    assert len(more) >= 0 and all(np.isfinite(more))
