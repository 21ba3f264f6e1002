import sys, os
os.environ.setdefault("KOFFT_HIP_HOST_PIPELINE", "0")  # one launch per call on the batch the case names (tests/conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.chdir(ROOT)
import numpy as np
import test_gpu_fuzz as F
import kofft_amd
from oracle import pyoracle as oracle
oracle.build()
f32 = kofft_amd.HipFftImpl(np.float32); f64 = kofft_amd.HipFftImpl(np.float64)
import conftest
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 160):
    for fn, args in ((F.test_fuzz_complex.__wrapped__ if hasattr(F.test_fuzz_complex,'__wrapped__') else F.test_fuzz_complex, (f32, f64, oracle, seed)),
                     (F.test_fuzz_real, (f32, f64, oracle, seed)), (F.test_fuzz_stft, (f32, oracle, seed))):
        try:
            fn(*args)
        except AssertionError as e:
            bad += 1
            print("FAIL", fn.__name__, seed, e, flush=True)
print("soak done, failures:", bad)
