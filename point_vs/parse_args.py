"""Command line of `point_vs.py`, flag for flag compatible with the reference's
(/root/reference/point_vs/parse_args.py:6-236): same names, short forms, types and defaults, so a
reference command line (or a `cmd_args.yaml` written by it, `--load_args`) parses unchanged.

The flags are declared as data (FLAGS below), not copied argparse calls. Flags that only steer parts
of PointVS outside the EGNN hot path (wandb, parquet preprocessing switches, the lucid/LieConv
models) are accepted and recorded so that existing command lines keep working; the ones that this
entry cannot honour are listed by `unsupported_in_use`.
"""
import argparse

STR, INT, FLT, ON = str, int, float, 'store_true'

# (names, type or ON, default, help)
FLAGS = (
    # --- data ---
    (('--train_data_root_pose',), STR, None, 'root of the pose-classification training structures'),
    (('--train_data_root_affinity', '--tdra'), STR, None, 'root of the affinity training structures'),
    (('--test_data_root_pose',), STR, None, 'root of the pose-classification validation structures'),
    (('--test_data_root_affinity',), STR, None, 'root of the affinity validation structures'),
    (('--train_types_pose',), STR, None, 'types file listing label / receptor / ligand per pose sample'),
    (('--train_types_affinity',), STR, None, 'types file of the affinity training set'),
    (('--test_types_pose',), STR, None, 'types file of the pose validation set'),
    (('--test_types_affinity',), STR, None, 'types file of the affinity validation set'),
    (('--translated_actives',), STR, None, 'directory of translated actives (recorded only)'),
    (('--input_suffix', '-s'), STR, 'parquet', 'file name suffix of the structure files'),
    (('--batch_size', '-b'), INT, 32, 'graphs per batch'),
    (('--radius',), INT, 10, 'box radius around the ligand centre, Angstrom'),
    (('--edge_radius',), FLT, 4.0, 'radius of the edge graph, Angstrom'),
    (('--estimate_bonds',), ON, False, 'intra-molecular edges only below 2 A (covalent estimate)'),
    (('--prune',), ON, False, 'drop receptor atoms not connected to the ligand'),
    (('--compact',), ON, False, 'one bit for ligand/receptor instead of doubled atom types'),
    (('--use_atomic_numbers',), ON, False, 'atomic-number features instead of smina types'),
    (('--hydrogens',), ON, False, 'keep polar hydrogens'),
    (('--extended_atom_types',), ON, False, '28-type scheme'),
    (('--augmented_actives',), INT, 0, 'randomly rotated copies of each active'),
    (('--min_aug_angle',), FLT, 30, 'minimum rotation of an augmented active, degrees'),
    (('--max_active_rmsd',), FLT, None, 'label threshold for actives'),
    (('--min_inactive_rmsd',), FLT, None, 'label threshold for inactives'),
    (('--max_inactive_rmsd',), FLT, None, 'discard inactives beyond this RMSD'),
    (('--p_remove_entity',), FLT, 0, 'probability of training on ligand or receptor alone'),
    (('--p_noise',), FLT, -1, 'probability of label noise'),
    (('--include_strain_info',), ON, False, 'append conformer strain to the features'),
    (('--synth_pharm', '-p'), ON, False, 'synthetic pharmacophore data set'),
    (('--synthpharm',), ON, False, 'synthetic pharmacophore data set (alias)'),
    # --- model ---
    (('--channels', '-k'), INT, 32, 'width of the node feature vectors'),
    (('--layers',), INT, 6, 'number of EGNN layers'),
    (('--activation',), STR, 'relu', 'recorded; the EGNN layers always use SiLU, as in the reference'),
    (('--dropout',), FLT, 0.0, 'edge dropout probability'),
    (('--egnn_attention',), ON, False, 'edge attention gate'),
    (('--egnn_tanh',), ON, False, 'tanh on the coordinate scalar'),
    (('--egnn_normalise',), ON, False, 'normalise coordinate differences'),
    (('--egnn_residual',), ON, False, 'residual node update'),
    (('--egnn_edge_residual',), ON, False, 'residual edge messages'),
    (('--graphnorm',), ON, False, 'GraphNorm in the node MLP'),
    (('--multi_fc',), ON, False, 'three-layer head'),
    (('--static_coords',), ON, False, 'do not update coordinates'),
    (('--permutation_invariance',), ON, False, 'edge MLP on h_i + h_j'),
    (('--node_attention',), ON, False, 'node attention gate'),
    (('--attention_activation_function',), STR, 'sigmoid', 'sigmoid | relu | silu | tanh'),
    (('--gated_residual',), ON, False, 'learned residual gate'),
    (('--rezero',), ON, False, 'ReZero residuals'),
    (('--softmax_attention',), ON, False, 'softmax over the incoming edges of a node'),
    (('--final_softplus',), ON, False, 'softplus on the regression output'),
    (('--multi_target_affinity',), ON, False, 'three regression targets'),
    (('--model_task',), STR, 'classification', 'classification | regression | both | multi_regression'),
    (('--fourier_features',), INT, 0, 'lucid model only (recorded)'),
    (('--norm_coords',), ON, False, 'lucid model only (recorded)'),
    (('--norm_feats',), ON, False, 'lucid model only (recorded)'),
    (('--thin_mlps',), ON, False, 'lucid model only (recorded)'),
    (('--lucid_node_final_act',), ON, False, 'lucid model only (recorded)'),
    # --- optimisation ---
    (('--epochs_pose', '-ep'), INT, 0, 'epochs over the pose training set'),
    (('--epochs_affinity', '-ea'), INT, 0, 'epochs over the affinity training set'),
    (('--learning_rate', '-lr'), FLT, 0.002, 'learning rate'),
    (('--weight_decay', '-w'), FLT, 1e-4, 'L2 weight decay'),
    (('--optimiser', '-o'), STR, 'adam', 'adam | sgd'),
    (('--regression_loss',), STR, 'mse', 'mse | huber'),
    (('--use_1cycle',), ON, False, '1cycle learning-rate schedule'),
    (('--warm_restarts',), ON, False, 'cosine annealing with warm restarts'),
    (('--double',), ON, False, 'fp64 (not available on the HIP path: raises)'),
    # --- run control / bookkeeping ---
    (('--load_weights', '-l'), STR, None, 'checkpoint (or model directory) to start from'),
    (('--load_args',), STR, None, 'yaml of argument values that override the command line'),
    (('--logging_level',), STR, 'info', 'notset | debug | info | warning | error | critical'),
    (('--wandb_project',), STR, None, 'recorded; wandb is not used by this entry'),
    (('--wandb_run',), STR, None, 'recorded; with --wandb_project selects save_path/project/run'),
    (('--wandb_dir',), STR, None, 'recorded'),
    (('--val_on_epoch_end', '-v'), ON, False, 'validate after every epoch'),
    (('--top1',), ON, False, 'recorded (top-1 analysis is outside the path)'),
    (('--only_save_best_models',), ON, False, 'keep only improving checkpoints'),
    (('--end_flag',), ON, False, 'write save_path/_FINISHED on completion'),
)

# extensions of this entry (not in the reference): data sources that need no parquet parsing
EXTRA_FLAGS = (
    (('--synthetic_graphs',), INT, 0, 'train / validate on this many synthetic radius graphs '
                                      '(SURVEY.md §8d generator) instead of a data root'),
    (('--synthetic_atoms',), INT, 500, 'atoms per synthetic graph'),
)


def build_parser():
    parser = argparse.ArgumentParser(
        description='PointVS EGNN training / validation on the MI355X-native path')
    parser.add_argument('model', type=str, help='egnn | multitask (lucid is outside the HIP path)')
    parser.add_argument('save_path', type=str, help='directory for checkpoints, predictions and yaml records')
    for names, kind, default, text in FLAGS + EXTRA_FLAGS:
        if kind == ON:
            parser.add_argument(*names, action='store_true', help=text)
        else:
            parser.add_argument(*names, type=kind, default=default, help=text)
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def unsupported_in_use(args):
    """Flags set on the command line that this entry cannot honour (it raises on them rather than
    silently training something else)."""
    bad = []
    if args.double:
        bad.append('--double (the HIP kernels are fp32)')
    if args.model == 'lucid':
        bad.append('model lucid (outside the hot path)')
    if args.include_strain_info:
        bad.append('--include_strain_info')
    return bad
