#!/bin/bash
# r06_seed.sh -- seed 520015 of the shard-only draw failed ONCE on the release build (step 5, 1.4e-5) and passed before and after: how often?
O=gpurun_out/r06seed; mkdir -p $O; : > $O/loop.txt
for i in $(seq 1 ${LOOPS:-45}); do
  FUZZ_RELEASE=1 FUZZ_SHARDS=1 timeout 120 python profiles/fuzz_parity.py 0.5 520015 2>&1 | grep -E "FAIL|fuzz ok" | cut -c1-260 >> $O/loop.txt
done
grep -c "fuzz ok" $O/loop.txt; grep FAIL $O/loop.txt | head
# the first 20 seeds of that leg, a few times over
for i in 1 2 3 4; do FUZZ_RELEASE=1 FUZZ_SHARDS=1 timeout 200 python profiles/fuzz_parity.py 30 520001 2>&1 | grep -E "FAIL|fuzz ok" | cut -c1-260 >> $O/leg.txt; done
cat $O/leg.txt
