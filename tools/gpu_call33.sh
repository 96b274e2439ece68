#!/bin/bash
O=gpurun_out/c33; mkdir -p $O
for v in spf0 spf1 spf0 spf1; do echo "== $v"; timeout 30 tools/_build/osblock_prof_$v 16384 5 x; done > $O/stem_prefetch_ab.txt 2>&1
cat $O/stem_prefetch_ab.txt
timeout 60 python -m pytest tests/test_gpu_reid.py -q -x -k "crops_bit_exact or reference_init_within_tolerance or multistream_fused or in_the_loop" > $O/pytest_reid.log 2>&1; echo "pytest rc=$?" >> $O/pytest_reid.log; tail -n 3 $O/pytest_reid.log | cut -c1-160
