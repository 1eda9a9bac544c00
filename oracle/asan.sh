#!/bin/bash
# The CPU oracle -- the only pin the kernels have -- under AddressSanitizer + UBSan. CPU only.
#   oracle/asan.sh            tests/test_oracle_cpu.py, tests/test_gltf_cpu.py + the golden regeneration against the sanitizer build
set -e
cd "$(dirname "$0")/.."
make -C oracle -s asan
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export PSM_ORACLE_LIB=$PWD/oracle/libpsm_oracle_asan.so
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
python -m pytest tests/test_oracle_cpu.py tests/test_gltf_cpu.py -q -x -p no:cacheprovider "$@"
# the golden fixtures again, into a scratch directory, and compared with the committed ones
out=$(mktemp -d)
PSM_GOLDEN_OUT=$out python tests/golden/make_golden.py
python - "$out" <<'PY'
import sys, os, numpy as np
out = sys.argv[1]
for f in sorted(os.listdir(out)):
    a, b = np.load(os.path.join(out, f)), np.load(os.path.join("tests/golden", f))
    assert set(a.files) == set(b.files), f
    for k in a.files:
        assert a[k].tobytes() == b[k].tobytes(), (f, k)
    print("golden", f, "identical under the sanitizer build:", len(a.files), "arrays")
PY
rm -rf "$out"
