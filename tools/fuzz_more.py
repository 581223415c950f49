"""Extra seeds of the kernel-family fuzz test (tests/test_gpu_properties.py); run with
PYTORCH_NO_CUDA_MEMORY_CACHING=1 to turn out-of-bounds accesses into faults.
Usage: python tools/fuzz_more.py [first_seed] [n_seeds]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'tests'))
sys.path.insert(0, str(ROOT))
import test_gpu_properties as t  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 8
count = int(sys.argv[2]) if len(sys.argv) > 2 else 32
bad = 0
for seed in range(first, first + count):
    try:
        t.test_kernel_families_agree_on_random_configurations(seed)
    except AssertionError as e:
        bad += 1
        print('FAIL', seed, str(e)[:300], flush=True)
print('done, failures:', bad)
