#!/usr/bin/env python3
"""Predictions files written by the REFERENCE's own `val()` loop (build container only):
/root/reference/point_vs/models/point_neural_network_base.py:208-360 on the first 7 samples of
data/small_chembl_test (pose, 3 + 3 + 1 per batch) and of data/multi_classification_sample
(affinity), random-init MultitaskSatorrasEGNN under seed 2 with the README flags. Stored with the
model's raw outputs and labels per batch in tests/golden/predictions_reference.json, so the
line formatting and file cadence of pointvs_amd/predictions.py can be pinned without a model."""
import importlib.util
import json
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.argv = ['x']
spec = importlib.util.spec_from_file_location('mg', HERE / 'make_golden.py')
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
import torch  # noqa: E402

README_KW = {'act': 'relu', 'bn': True, 'cache': False, 'ds_frac': 1.0, 'k': 32, 'num_layers': 3, 'dropout': 0.0,
             'dim_input': 22, 'dim_output': 1, 'edge_attention': False, 'tanh': False, 'normalize': False,
             'residual': False, 'edge_residual': False, 'graphnorm': False, 'multi_fc': False, 'update_coords': True,
             'permutation_invariance': False, 'attention_activation_fn': 'sigmoid', 'node_attention': False,
             'gated_residual': False, 'rezero': False, 'model_task': 'classification', 'final_softplus': False,
             'softmax_attention': False}
out = {}
for tag, root, types, task in (('pose', 'data/small_chembl_test', 'data/small_chembl_test.types', 'classification'),
                               ('affinity', 'data/multi_classification_sample',
                                'data/multi_classification_sample.types', 'regression')):
    with tempfile.TemporaryDirectory() as tmp:
        lines = Path(types).read_text().splitlines()
        pick = lines[:4] + lines[-3:]            # decoys and actives for the pose set
        short = Path(tmp) / 'short.types'
        short.write_text('\n'.join(pick) + '\n')
        dl = mg.get_data_loader(
            Path(root), mg.PygPointCloudDataset, types_fname=short, mode='val', model_task=task, batch_size=3,
            compact=False, radius=10, use_atomic_numbers=False, rot=False, polar_hydrogens=False,
            fname_suffix='parquet', edge_radius=4.0, estimate_bonds=False, prune=False,
            extended_atom_types=False, include_strain_info=False)
        torch.manual_seed(2)
        model = mg.MultitaskSatorrasEGNN(Path(tmp) / 'run', 2e-3, 1e-4, None, None, silent=False, **README_KW)
        model.set_task(task)
        model.log_interval = 2                  # exercises the periodic flush (:492-499) on 3 batches
        batches = []
        with torch.no_grad():
            for g in dl:
                y_pred, y_true, ligs, recs = model.unpack_input_data_and_predict(mg.clone_graph(g))
                batches.append({'y_pred': [float(v) for v in y_pred.reshape(-1)],
                                'y_true': [float(v) for v in y_true.reshape(-1)],
                                'ligands': [str(p) for p in ligs], 'receptors': [str(p) for p in recs]})
        model.val(dl)
        fname = Path(tmp) / 'run' / f'{"pose" if task == "classification" else "affinity"}_predictions.txt'
        out[tag] = {'task': task, 'file_name': fname.name, 'text': fname.read_text(), 'batches': batches}
(HERE / 'predictions_reference.json').write_text(json.dumps(out, indent=1) + '\n')
print({k: v['text'].count('\n') for k, v in out.items()})
print(out['pose']['text'])
