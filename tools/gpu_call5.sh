#!/bin/bash
O=gpurun_out/c5; mkdir -p $O
for v in base dense base dense; do echo "== $v"; timeout 120 tools/_build/osblock_prof_$v 4096 8 x | grep -E "best"; timeout 120 tools/_build/osblock_prof_$v 4096 8 | grep -E "best|weighted" | head -8; done > $O/osblock_dense.txt 2>&1
BOXMOT_HIP_LIB=$PWD/tools/_build/libboxmot_hip_dense.so timeout 300 python -m pytest tests/test_gpu_reid.py -q -m gpu -x > $O/pytest_dense.log 2>&1
echo "rc=$?" >> $O/pytest_dense.log
timeout 600 python tools/ab_variants.py run base dense > $O/ab.txt 2>&1
cat $O/osblock_dense.txt; tail -n 3 $O/pytest_dense.log; cat $O/ab.txt
