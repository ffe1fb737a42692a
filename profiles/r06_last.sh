#!/bin/bash
# r06_last.sh -- the library as committed last: smoke, the step-path tests with the HDP write-back on, then the default once
O=gpurun_out/r06last; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
HC_HDP_FLUSH=1 timeout 600 python -m pytest tests/test_gpu_boundary.py tests/test_chrono_adapter.py tests/test_gpu_ahead.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_boundary.py tests/test_chrono_adapter.py tests/test_capi_exports.py -x -q 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | cut -c1-200
