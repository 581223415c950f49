"""Boundary pieces around the path that need no GPU: the reference CLI surface of point_vs.py /
point_vs.parse_args, the reference import paths (point_vs.models.geometric.*), and the predictions
file the validation loop writes, pinned on a file written by the reference's own val()."""
import argparse
import importlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

GOLDEN = Path(__file__).resolve().parent / 'golden'
ROOT = Path(__file__).resolve().parent.parent


def test_cli_flags_match_the_reference_parser():
    """Every option string, type, default and store_true switch of
    /root/reference/point_vs/parse_args.py:6-236 (captured as data by make_golden_cli.py)."""
    from point_vs.parse_args import build_parser
    ref = json.loads((GOLDEN / 'cli_flags.json').read_text())['flags']
    mine = {a.dest: a for a in build_parser()._actions if not isinstance(a, argparse._HelpAction)}
    assert len(ref) == 76
    for f in ref:
        a = mine[f['dest']]
        assert sorted(list(a.option_strings) or [a.dest]) == sorted(f['names']), f['dest']
        assert a.default == f['default'], f['dest']
        assert (None if a.type is None else a.type.__name__) == f['type'], f['dest']
        assert isinstance(a, argparse._StoreTrueAction) == f['store_true'], f['dest']
    assert set(mine) - {f['dest'] for f in ref} == {'synthetic_graphs', 'synthetic_atoms'}


def test_readme_command_line_maps_to_the_reference_model_kwargs():
    """The README example's flags -> build_net kwargs as point_vs.py:189-221 derives them; the
    expected dict is the `kwargs` recorded in the config-1 golden (made from the reference)."""
    import point_vs as entry_pkg   # the package (directory) shadows the script, as in the reference
    spec = importlib.util.spec_from_file_location('pvs_entry', Path(entry_pkg.__file__).parents[1] / 'point_vs.py')
    entry = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(entry)
    from point_vs.parse_args import parse_args
    args = parse_args(['multitask', '/tmp/x', '--model_task', 'both', '-ea', '1', '-ep', '1', '--layers', '3'])
    kw = entry.model_kwargs_from_args(args, 22, 'regression')
    kw['model_task'] = 'classification'         # point_vs.py:223-224 for --model_task both
    want = json.loads(str(np.load(GOLDEN / 'c5_config1_pose_real3.npz')['cfg']))['kwargs']
    assert {k: kw[k] for k in want} == want


def test_reference_import_paths_resolve_to_the_hip_classes():
    import pointvs_amd.egnn_multitask as mt
    import pointvs_amd.egnn_satorras as sat
    assert importlib.import_module('point_vs.models.geometric.egnn_satorras').EGNNLayer is sat.EGNNLayer
    assert importlib.import_module('point_vs.models.geometric.egnn_satorras').SartorrasEGNN is sat.SartorrasEGNN
    assert importlib.import_module('point_vs.models.geometric.egnn_multitask').MultitaskSatorrasEGNN \
        is mt.MultitaskSatorrasEGNN
    base = importlib.import_module('point_vs.models.geometric.pnn_geometric_base')
    assert base.PygLinearPass.__module__ == 'pointvs_amd.pnn_geometric_base'
    assert importlib.import_module('point_vs.models.point_neural_network_base').PointNeuralNetworkBase
    assert importlib.import_module('point_vs.global_objects').DEVICE is not None


def test_predictions_writer_reproduces_the_reference_file(tmp_path):
    """Feeding the reference model's raw outputs through the streaming writer gives the file the
    reference's val() wrote, byte for byte (line format :287-325, periodic append :492-499)."""
    from pointvs_amd.predictions import PredictionsWriter
    ref = json.loads((GOLDEN / 'predictions_reference.json').read_text())
    for tag, rec in ref.items():
        path = tmp_path / rec['file_name']
        with PredictionsWriter(path, rec['task'], flush_every=2) as w:
            for b in rec['batches']:
                y = torch.tensor(b['y_pred'])
                w.submit(torch.sigmoid(y) if rec['task'] == 'classification' else y, torch.tensor(b['y_true']),
                         b['receptors'], b['ligands'])
        assert path.read_text() == rec['text'], tag
        assert w.lines_written == 7


def test_prediction_line_formats_without_labels_and_multi_target():
    from pointvs_amd.predictions import format_lines
    assert format_lines('classification', [0.25], None, ['r'], ['l']) == ['0.250 | r l']
    assert format_lines('multi_regression', [[1, 2, 3]], None, ['r'], ['l']) == ['1.000 2.000 3.000 | r l']
    got = format_lines('multi_regression', [[1, 2, 3], [4, 5, 6]], [[-1, 7, -1], [8, -1, -1]], ['ra', 'rb'], ['la', 'lb'])
    assert got == ['7.000 | 2.000 ra la | pkd', '8.000 | 4.000 rb lb | pki']


def test_bench_launcher_parent_never_loads_torch_and_counts_gpus_from_sysfs(tmp_path, monkeypatch):
    """`python bench.py --gpus N` starts its ranks from a parent that has not even
    imported torch (so it cannot have initialised HIP); the GPU count comes from the KFD topology."""
    import subprocess
    code = ("import sys, bench; assert 'torch' not in sys.modules, 'torch imported by the launcher'; "
            "assert not any('libamdhip64' in l for l in open('/proc/self/maps')); print(bench.visible_gpu_count())")
    out = subprocess.run([sys.executable, '-c', code], cwd=str(ROOT), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    import bench
    for node, simd in enumerate((0, 0, 1024, 1024, 1024)):           # two CPU agents, three GPUs
        d = tmp_path / str(node)
        d.mkdir()
        (d / 'properties').write_text(f'cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n')
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count(tmp_path) == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,2')
    assert bench.visible_gpu_count(tmp_path) == 2
    assert bench.visible_gpu_count(tmp_path / 'absent') is None


def test_bench_eight_rank_launch_command_device_binding_and_rccl_choice(monkeypatch):
    """the first real 8-GPU session must not fail on trivia. `python bench.py --gpus 8
    --global-batch 256 --rccl-proto LL` (BASELINE config 4): the parent builds ONE torch.distributed.run command with
    eight ranks on the loopback that re-runs bench.py with the same arguments, hands NCCL_PROTO (and the dmabuf IPC
    switch) to the ranks' environment before any process group exists, and refuses when fewer GPUs are visible than
    ranks - unless the dry-run backend is named. A rank binds LOCAL_RANK -> its own device (checked on the source:
    no GPU here)."""
    import argparse
    import bench
    args = argparse.Namespace(gpus=8, rccl_proto='LL', rccl_algo=None)
    argv = ['--gpus', '8', '--global-batch', '256', '--rccl-proto', 'LL']
    cmd = bench.launch_command(args, 29517, argv)
    assert cmd[:4] == [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1']
    assert '--nproc-per-node=8' in cmd and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29517'
    assert cmd[-len(argv) - 1:] == [str(ROOT / 'bench.py')] + argv
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env):
        seen['cmd'], seen['env'] = cmd, env
        return Done()
    monkeypatch.delenv('PVS_BENCH_BACKEND', raising=False)
    monkeypatch.delenv('NCCL_PROTO', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py'] + argv)
    assert bench.self_launch(args, run=fake_run, n_visible=8) == 0
    assert seen['env']['NCCL_PROTO'] == 'LL' and seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert 'NCCL_ALGO' not in seen['env'] and seen['cmd'][-len(argv):] == argv
    assert bench.rccl_env_record(seen['env'])['NCCL_PROTO'] == 'LL'
    with pytest.raises(SystemExit, match='only 1 GPU'):
        bench.self_launch(args, run=fake_run, n_visible=1)
    monkeypatch.setenv('PVS_BENCH_BACKEND', 'gloo')
    assert bench.self_launch(args, run=fake_run, n_visible=1) == 0           # the dry run shares devices
    src = (ROOT / 'bench.py').read_text()
    assert "local_rank = int(os.environ.get('LOCAL_RANK', 0))" in src
    assert "dev_index = local_rank if backend == 'nccl' else" in src and 'torch.cuda.set_device(dev_index)' in src
    assert src.index('apply_rccl_choice(args, os.environ)') < src.index("dist.init_process_group('nccl'")


def test_model_selection_metrics_follow_the_reference(tmp_path):
    """val()'s top1_on_end: top-n over receptors (analysis/top_n.py:32-49) and Pearson's r of the affinity file
    (utils.py:189-198) on predictions files in the reference's line format."""
    from scipy.stats import pearsonr
    from pointvs_amd.predictions import regression_pearson, top_n
    pose = tmp_path / 'pose_predictions.txt'
    pose.write_text('\n'.join([
        '1.000 | 0.900 recA ligA_0', '0.000 | 0.950 recA ligA_1', '0.000 | 0.100 recA ligA_2',     # best pose is a decoy
        '1.000 | 0.700 recB ligB_0', '0.000 | 0.200 recB ligB_1',                                 # best pose is active
        '0.000 | 0.300 recC ligC_0', '0.000 | 0.600 recC ligC_1']) + '\n')                        # no active at all
    assert top_n(pose) == pytest.approx(1 / 3) and top_n(pose, n=2) == pytest.approx(2 / 3)
    aff = tmp_path / 'affinity_predictions.txt'
    y = np.array([4.1, 5.0, 6.2, 7.7, 5.5, 8.1]); yp = np.array([4.5, 4.9, 6.0, 7.0, 6.1, 7.9])
    aff.write_text('\n'.join(f'{a:.3f} | {b:.3f} rec lig{i}' for i, (a, b) in enumerate(zip(y, yp))) + '\n')
    r, p = regression_pearson(aff)
    r0, p0 = pearsonr(y, yp)
    assert r == pytest.approx(r0, abs=1e-3) and p < 0.05 and p0 < 0.05


def test_bench_limiter_label_follows_the_parsed_shares(tmp_path, monkeypatch):
    """bench.py's roofline.limiter is derived from the shares of the newest SQ counter summary (every
    resource over its threshold - VALU or matrix pipe >= 50 % busy, a wave >= 30 % parked in s_waitcnt - else the largest
    share), not a constant; the file's first line names the commit it was measured at."""
    import bench
    prof = tmp_path / 'profiles'
    prof.mkdir()
    line = ('  {k}: kernel 1e+06 cycles per launch; VALU busy {v}% of SIMD time, matrix pipe {m}% (10% of it under VALU work); '
            'per wave: issuing 40%, issue-stalled {st}%, in s_waitcnt {w}%; LDS busy 20% of CU time (0% of it bank conflicts)')
    (prof / 'r07_cfg2_pmc_sq.txt').write_text('commit: abc1234\nheader\n' + line.format(k='k_a<0>', v=58, m=18, st=20, w=37) + '\n' +
                                              line.format(k='k_b<1>', v=82, m=11, st=39, w=24) + '\n' +
                                              line.format(k='k_c', v=30, m=62, st=5, w=12) + '\n' +
                                              line.format(k='k_d', v=20, m=10, st=5, w=25) + '\n')
    (prof / 'r06_cfg2_pmc_sq.txt').write_text(line.format(k='k_a<0>', v=99, m=99, st=0, w=0) + '\n')     # older: ignored
    monkeypatch.setattr(bench, 'ROOT', tmp_path)
    a = bench.measured_limiter('cfg2', 'k_a')
    assert a['limiter'] == 'valu_issue+wave_stalls' and a['valu_busy'] == 0.58 and a['waitcnt_share'] == 0.37
    assert a['limiter_source'] == 'r07_cfg2_pmc_sq.txt' and a['limiter_commit'] == 'abc1234'
    assert bench.measured_limiter('cfg2', 'k_b')['limiter'] == 'valu_issue'
    assert bench.measured_limiter('cfg2', 'k_c')['limiter'] == 'matrix_pipe'
    assert bench.measured_limiter('cfg2', 'k_d')['limiter'] == 'wave_stalls'         # nothing over a threshold: the largest share
    assert bench.measured_limiter('cfg9', 'k_a')['limiter'] is None
