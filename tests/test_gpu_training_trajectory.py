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
    # every checkpoint - also those written at epoch ends WHILE the replayer had the optimiser in its capturable form -
    # holds the optimiser as the reference's checkpoints do: not capturable, step counters on the host
    for rel_path in ckpts:
        raw = torch.load(tmp_path / rel_path, weights_only=False)        # (no map_location: devices as saved)
        osd = raw['optimiser_state_dict']
        assert not any(g.get('capturable') for g in osd['param_groups']), rel_path
        assert all(not st['step'].is_cuda for st in osd['state'].values()), rel_path
        assert all(st['exp_avg'].shape == st['exp_avg_sq'].shape for st in osd['state'].values())
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



@pytest.mark.parametrize('name,legacy', [('default', False), ('k64_attention', False), ('k64_attention', True)])
def test_resuming_from_a_checkpoint_the_reference_wrote(name, legacy, tmp_path):
    """SURVEY 8f row 4 pinned on the REFERENCE's output: tests/golden/ckpt_<name>.pt is the file the
    reference's save() (:501-517) wrote after the twelve steps of the `<name>` trajectory, byte for byte;
    ckpt_k64_attention_legacy.pt carries the older names (`edge_attention_mlp.2.*`, `node_attention_mlp.*`) and was
    kept only after the reference's own load_weights had loaded it (tests/golden/make_golden_training.py). Here a model
    with OTHER initial weights loads the file and must then
      * hold the file's weights and epoch counters, and give the logits the resumed REFERENCE model gave (1e-5);
      * hold the file's optimiser state parameter by parameter (step = 12, both moments): the reference's Adam keys its
        state by the position of a parameter in `model.parameters()`, so this pins the registration order too;
      * take the run's 13th optimiser step like the resumed reference did: the loss of `backprop()` on the first batch,
        the learning rate it ran at, and the weights it leaves."""
    import shutil
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    z, meta = _load(name)
    torch.manual_seed(meta['seed'] + 1000)
    model = SartorrasEGNN(tmp_path, meta['lr'], meta['wd'], None, None, silent=True, **meta['ctor'], **meta['kwargs'])
    src = GOLDEN / (f'ckpt_{name}_legacy.pt' if legacy else f'ckpt_{name}.pt')
    dst = tmp_path / 'checkpoints' / 'pose_ckpt_epoch_3.pt'
    dst.parent.mkdir(parents=True, exist_ok=True)
    shutil.copyfile(src, dst)
    file = torch.load(GOLDEN / f'ckpt_{name}.pt', map_location='cpu', weights_only=True)     # (plain names)
    assert any(not np.array_equal(v.detach().cpu().numpy(), file['model_state_dict'][k].numpy())
               for k, v in model.state_dict().items())
    model.load_weights(dst, silent=True)
    assert (model.p_epoch, model.a_epoch) == (int(z['resume/p_epoch']), int(z['resume/a_epoch'])) == (3, 0)
    sd = model.state_dict()
    assert list(sd.keys()) == list(file['model_state_dict'].keys())
    for k, v in sd.items():
        assert np.array_equal(v.detach().cpu().numpy(), file['model_state_dict'][k].numpy()), k
    # optimiser state, by position in the reference's parameter order (= its state_dict order: no buffers here)
    names = [n for n, _ in model.named_parameters()]
    assert names == list(file['model_state_dict'].keys())
    fstate = file['optimiser_state_dict']['state']
    restored = 0
    for i, (n, p) in enumerate(model.named_parameters()):
        st = model.optimiser.state.get(p)
        if i not in fstate:                    # (the last layer's coord_mlp never had a gradient: SURVEY Q3)
            assert not st, n
            continue
        assert float(st['step']) == 12.0 and not st['step'].is_cuda, n
        assert torch.equal(st['exp_avg'].cpu(), fstate[i]['exp_avg']), n
        assert torch.equal(st['exp_avg_sq'].cpu(), fstate[i]['exp_avg_sq']), n
        restored += 1
    assert restored == len(fstate) and restored >= len(names) - 3

    first = _loaders(z, meta)[0][0].to('cuda')
    model.eval()
    with torch.no_grad():
        logits = model(first).reshape(-1).cpu().numpy()
    ref_logits = z['resume/logits']
    assert np.abs(logits - ref_logits).max() <= 1e-5 * max(1.0, np.abs(ref_logits).max()), (logits, ref_logits)
    model.train()
    assert float(model.optimiser.param_groups[0]['lr']) == float(z['resume/lr'])
    y_pred, y_true, _, _ = model.unpack_input_data_and_predict(first)
    loss = float(model.backprop(y_true, y_pred))
    assert abs(loss - float(z['resume/loss'])) <= 1e-5 * abs(float(z['resume/loss'])), (loss, float(z['resume/loss']))
    steps = sorted({int(st['step']) for st in model.optimiser.state.values() if st})
    assert steps == z['resume/optimiser_steps'].tolist() == [13]
    # the resumed run takes the FUSED optimiser step (the file was read with map_location = the device, which brings
    # the step counters in as device tensors: load_state_dict puts them back on the host, where this flavour counts)
    assert model.optimiser._fast is not None and model.optimiser._fast['fusable']
    for k, v in model.state_dict().items():
        got, ref = v.detach().cpu().numpy().astype(np.float64), z[f'sd13/{k}'].astype(np.float64)
        before = file['model_state_dict'][k].numpy().astype(np.float64)
        d, moved = np.abs(got - ref), np.abs(ref - before)
        # one Adam step moves an entry by at most ~lr; the step itself must be the reference's to a few percent of its
        # size (an entry whose moments are rounding noise can land anywhere within lr)
        assert d.max() <= 2e-3 * 1.001, k
        assert np.median(d) <= 2e-2 * float(np.median(moved)) + 1e-9, (k, float(np.median(d)), float(np.median(moved)))
