#!/usr/bin/env python3
"""`point_vs.py` entry of the reference (/root/reference/point_vs.py:36-275) on the MI355X-native
EGNN path: same positional arguments and flags (point_vs/parse_args.py), same mapping from flags to
model kwargs (:189-221), same order of work (pose training -> pose validation -> affinity training
-> affinity validation, :258-275), same records in save_path (cmd_args.yaml, model_kwargs.yaml,
checkpoints/, predictions files).

    python point_vs.py multitask /tmp/run --model_task both -ea 1 -ep 1 --layers 3 \\
        --train_data_root_pose graphs/pose --train_data_root_affinity graphs/affinity ...

Data sources. Turning parquet structure files into graphs is the reference's CPU data layer and is
outside the hot-path scope; this entry reads
  * a data root holding one `.npz` per graph: `x [N,F]`, `pos [N,3]`, `y`, and either
    `edge_index [2,E]` + `edge_type [E]` (as the reference's loader emits them; an optional
    `edge_layout = 'generate_edges'` entry promises that order and selects the sort-free preparation) or, without edges,
    the ligand/receptor bit in the last column of `x` - the radius graph is then built on the GPU
    with --edge_radius (pvs_radius_graph_*); a types file, when given, lists the files to use
    (last column = file name relative to the root, first column = label);
  * `--synthetic_graphs N`: N synthetic protein-ligand radius graphs (SURVEY.md §8d generator).
One process per GPU under `python -m torch.distributed.run`: ranks draw disjoint shares of one
seeded sample sequence (pointvs_amd/data_loaders.py) and exchange gradients over RCCL
(pointvs_amd/distributed.py).
"""
import os
import socket
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from point_vs.parse_args import parse_args, unsupported_in_use  # noqa: E402


def model_kwargs_from_args(args, dim_input, regression_task):
    """Flags -> build_net kwargs, the mapping of /root/reference/point_vs.py:189-221 (the layer
    switches are store_true flags, so a bare command line builds the all-off layer)."""
    return {
        'act': args.activation, 'bn': True, 'cache': False, 'ds_frac': 1.0,
        'k': args.channels, 'num_layers': args.layers, 'dropout': args.dropout,
        'dim_input': dim_input, 'dim_output': 3 if regression_task == 'multi_regression' else 1,
        'norm_coords': args.norm_coords, 'norm_feats': args.norm_feats, 'thin_mlps': args.thin_mlps,
        'edge_attention': args.egnn_attention, 'attention': args.egnn_attention,
        'tanh': args.egnn_tanh, 'normalize': args.egnn_normalise, 'residual': args.egnn_residual,
        'edge_residual': args.egnn_edge_residual, 'graphnorm': args.graphnorm,
        'multi_fc': args.multi_fc, 'update_coords': not args.static_coords,
        'node_final_act': args.lucid_node_final_act,
        'permutation_invariance': args.permutation_invariance,
        'attention_activation_fn': args.attention_activation_function,
        'node_attention': args.node_attention, 'gated_residual': args.gated_residual,
        'rezero': args.rezero, 'model_task': args.model_task,
        'include_strain_info': args.include_strain_info, 'final_softplus': args.final_softplus,
        'softmax_attention': args.softmax_attention,
    }


class NpzGraphs:
    """Indexable data set over `.npz` graph files (see the module docstring for the format)."""

    def __init__(self, root, types_fname=None, task='classification', suffix='npz'):
        from pointvs_amd.graph import Data
        self._data_cls = Data
        self.root, self.task = Path(root).expanduser(), task
        if types_fname is not None:
            rows = [ln.split() for ln in Path(types_fname).expanduser().read_text().splitlines() if ln.strip()]
            self.files = [self.root / Path(r[-1]).with_suffix('.' + suffix) for r in rows]
            col = 0 if task == 'classification' else min(1, len(rows[0]) - 2)
            self.labels = [float(r[col]) for r in rows]
        else:
            self.files = sorted(self.root.glob(f'**/*.{suffix}'))
            self.labels = None
        if not self.files:
            raise FileNotFoundError(f'no .{suffix} graphs under {self.root} (parquet structure files are '
                                    f'the reference\'s CPU data layer, outside this entry: see the docstring)')
        self.feature_dim = int(np.load(self.files[0])['x'].shape[1])

    def __len__(self):
        return len(self.files)

    def label(self, i):
        return float(np.load(self.files[i])['y']) if self.labels is None else self.labels[i]

    def __getitem__(self, i):
        z = np.load(self.files[i])
        y = torch.tensor(self.label(i))
        y = y.long() if self.task == 'classification' else y.float()
        item = dict(x=torch.from_numpy(z['x']).float(), pos=torch.from_numpy(z['pos']).float(), y=y,
                    lig_fname=str(self.files[i].name), rec_fname=str(self.files[i].parent.name))
        if 'edge_index' in z.files:
            item['edge_index'] = torch.from_numpy(z['edge_index'].astype(np.int64))
            item['edge_attr'] = torch.nn.functional.one_hot(torch.from_numpy(z['edge_type'].astype(np.int64)), 3)
            if 'edge_layout' in z.files:      # e.g. 'generate_edges': lists dumped from the reference loader
                item['edge_layout'] = str(z['edge_layout'])
        return self._data_cls(**item)


class SyntheticGraphs:
    def __init__(self, n, n_atoms, edge_radius, task, seed0):
        from pointvs_amd.synthetic import synthetic_graph
        self.items = [synthetic_graph(seed0 + k, n_nodes=n_atoms, n_lig=min(30, n_atoms // 4),
                                      edge_radius=edge_radius) for k in range(n)]
        if task != 'classification':
            for k, it in enumerate(self.items):
                it.y = torch.tensor(4.0 + (k % 7))
        self.feature_dim = int(self.items[0].x.shape[1])

    def __len__(self):
        return len(self.items)

    def label(self, i):
        return float(self.items[i].y)

    def __getitem__(self, i):
        return self.items[i]


class _WithRadiusGraph:
    """Loader wrapper: batches whose graphs carry no edge list get their radius graph built on the
    GPU from the coordinates (generate_edges semantics, preprocessing.py:68-155)."""

    def __init__(self, loader, args):
        self.loader, self.args = loader, args

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        from pointvs_amd.radius_graph import attach_radius_graph
        for batch in self.loader:
            if getattr(batch, 'edge_index', None) is None:
                r = self.args.edge_radius if self.args.edge_radius > 0 else 4.0
                attach_radius_graph(batch, r, 2.0 if self.args.estimate_bonds else r)
            yield batch


def make_loader(args, root, types_fname, mode, task, rank, world, seed0):
    from pointvs_amd.data_loaders import GraphLoader, RankWeightedSampler, class_balance_weights
    from pointvs_amd.distributed import shard_range
    from pointvs_amd.global_objects import DEVICE
    if args.synthetic_graphs:
        ds = SyntheticGraphs(args.synthetic_graphs, args.synthetic_atoms, args.edge_radius, task, seed0)
    elif root is not None:
        ds = NpzGraphs(root, types_fname, task, 'npz')
    else:
        return None
    if mode == 'train':
        weights = None
        if task == 'classification':     # class-balancing draw, data_loaders.py:170-186
            weights = class_balance_weights([int(ds.label(i)) for i in range(len(ds))])
        sampler = RankWeightedSampler(weights, len(ds), rank, world, seed=seed0)
    else:                                # validation: every rank scores its contiguous share, in order
        lo, hi = shard_range(len(ds), rank, world)
        sampler = list(range(lo, hi))
    loader = _WithRadiusGraph(GraphLoader(ds, args.batch_size, sampler=sampler, device=DEVICE), args)
    loader.dataset = ds
    loader.sampler = sampler
    return loader


def main(argv=None):
    args = parse_args(argv)
    if args.model_task == 'both' and args.model != 'multitask':
        raise RuntimeError('--model_task both (pose training followed by affinity training) needs the '
                           'multitask model')
    if args.load_args is not None:
        for key, value in yaml.safe_load(Path(args.load_args).expanduser().read_text()).items():
            if hasattr(args, key):
                setattr(args, key, value)
    problems = unsupported_in_use(args)
    if problems:
        raise NotImplementedError('not available on the HIP path: ' + '; '.join(problems))
    if args.wandb_project is not None and args.wandb_run is None:
        raise SystemExit('wandb_run must be specified if wandb_project is specified.')
    save_path = Path(args.save_path, *(p for p in (args.wandb_project, args.wandb_run) if p)).expanduser()

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
        dist.init_process_group(os.environ.get('PVS_BACKEND', 'nccl'))
    if rank == 0:
        save_path.mkdir(parents=True, exist_ok=True)
        args.hostname = socket.gethostname()
        args.slurm_jobid = os.getenv('SLURM_JOBID')
        (save_path / 'cmd_args.yaml').write_text(yaml.dump(vars(args)))

    from pointvs_amd.egnn_multitask import MultitaskSatorrasEGNN
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    model_class = {'egnn': SartorrasEGNN, 'multitask': MultitaskSatorrasEGNN}.get(args.model)
    if model_class is None:
        raise NotImplementedError('model must be one of multitask, egnn')
    regression_task = 'multi_regression' if (args.multi_target_affinity or
                                             args.model_task == 'multi_regression') else 'regression'

    want_pose = args.model_task != 'regression'
    want_aff = args.model_task in ('both', 'regression', 'multi_regression')
    train_pose = make_loader(args, args.train_data_root_pose, args.train_types_pose, 'train',
                             'classification', rank, world, 1000) if want_pose else None
    train_aff = make_loader(args, args.train_data_root_affinity, args.train_types_affinity, 'train',
                            regression_task, rank, world, 2000) if want_aff else None
    test_pose = make_loader(args, args.test_data_root_pose, args.test_types_pose, 'val',
                            'classification', rank, world, 3000) if 'regression' not in args.model_task else None
    test_aff = make_loader(args, args.test_data_root_affinity, args.test_types_affinity, 'val',
                           regression_task, rank, world, 4000) if args.model_task != 'classification' else None
    first = train_pose or train_aff or test_pose or test_aff
    if first is None:
        raise SystemExit('no data: give a data root of .npz graphs or --synthetic_graphs N')

    model_kwargs = model_kwargs_from_args(args, first.dataset.feature_dim, regression_task)
    if args.model_task == 'both':
        model_kwargs['model_task'] = 'classification'
    model = model_class(save_path, args.learning_rate, args.weight_decay, wandb_project=args.wandb_project,
                        use_1cycle=args.use_1cycle, warm_restarts=args.warm_restarts,
                        only_save_best_models=args.only_save_best_models,
                        regression_loss=args.regression_loss, optimiser=args.optimiser,
                        silent=rank != 0, **model_kwargs)
    if args.load_weights is not None:
        model.load_weights(args.load_weights)
    if world > 1:
        from pointvs_amd.distributed import OverlappedGradAllReducer
        model.grad_sync = OverlappedGradAllReducer(list(model.parameters()))

    if args.epochs_pose and train_pose is not None:
        model.set_task('classification')
        model.train_model(train_pose, epochs=args.epochs_pose, top1_on_end=args.top1,
                          epoch_end_validation_set=test_pose if args.val_on_epoch_end else None)
    if test_pose is not None:
        model.set_task('classification')
        model.val(test_pose, top1_on_end=args.top1)
    if args.epochs_affinity and train_aff is not None:
        model.set_task(regression_task)
        model.train_model(train_aff, epochs=args.epochs_affinity, top1_on_end=args.top1,
                          epoch_end_validation_set=test_aff if args.val_on_epoch_end else None)
    if test_aff is not None:
        model.set_task(regression_task)
        model.val(test_aff, top1_on_end=args.top1)
    if args.end_flag and rank == 0:
        (save_path / '_FINISHED').write_text('')
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return model


if __name__ == '__main__':
    main()
