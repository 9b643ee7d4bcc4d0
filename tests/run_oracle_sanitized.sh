#!/bin/bash
# CPU-only: build the oracle with AddressSanitizer + UBSan and run its CPU test files against that build.
# (GPU ASan / xnack runs are not available on this pool; the product's device code is covered by the parity tests.)
set -euo pipefail
cd "$(dirname "$0")/.."
gcc -O1 -g -fPIC -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -shared \
    -o /tmp/liboracle_asan.so oracle/g2o_graph_oracle.c oracle/localization_oracle.c -lm
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1
export ORACLE_SO_OVERRIDE=/tmp/liboracle_asan.so
python - <<'PY'
import os, sys
sys.path.insert(0, '.')
from oracle import oracle as O
O.build = lambda force=False: os.environ['ORACLE_SO_OVERRIDE']
import pytest
sys.exit(pytest.main(['-q', '-x', 'tests/test_oracle_kats.py', 'tests/test_oracle_frontend.py', 'tests/test_oracle_golden.py', '-p', 'no:cacheprovider']))
PY
