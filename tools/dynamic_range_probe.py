"""Row-relative errors of one EGNN layer on one graph whose node rows span 2^(+-R), per kernel family, against the fp64
oracle (tests/test_gpu_properties.py::dynamic_range_errors). Prints the table kept in profiles/rNN_dynamic_range.txt.
Usage (GPU box): python tools/dynamic_range_probe.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests.test_gpu_properties import dynamic_range_errors  # noqa: E402

ATT = dict(edge_attention=True, node_attention=True)
print('# max over rows (99th percentile) of  max_c|gpu - ref64| / max_c|ref64|,  every row relative to its own magnitude')
for hid in (32, 64):
    for name, flags in (('default', {}), ('attention', ATT)):
        for R in (0, 6, 12, 20):
            rec = dynamic_range_errors(hid, flags, R)
            for pname, err in rec['f16x2']['param_grads'].items():     # per tensor, relative to its own largest entry
                e32 = rec['fp32']['param_grads'][pname]
                print(f'H={hid:3d} {name:9s} range 2^+-{R:<2d} grad {pname:36s} f16x2 {err:9.2e}   fp32 family {e32:9.2e}   '
                      f'ratio {err / max(e32, 1e-30):7.2f}')
            for tensor in rec['f16x2']:
                if tensor == 'param_grads':
                    continue
                a, b = rec['f16x2'][tensor], rec['fp32'][tensor]
                print(f'H={hid:3d} {name:9s} range 2^+-{R:<2d} {tensor:15s} f16x2 {a[0]:9.2e} ({a[1]:8.2e})   '
                      f'fp32 family {b[0]:9.2e} ({b[1]:8.2e})   ratio {a[0] / max(b[0], 1e-30):7.2f}')
