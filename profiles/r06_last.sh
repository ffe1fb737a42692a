#!/bin/bash
# r06_last.sh -- the library as committed last (the sample-storing workgroup requests its canary word unconditionally): smoke, the
# step kernel's bitwise A/B, the sphere-decay golden through the C++ adapter, one driver line
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_boundary.py tests/test_chrono_adapter.py -x -q -m gpu -k "step_kernel_of_the_common or hdp_write_back or sphere_decay or state_behind" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | cut -c1-160
