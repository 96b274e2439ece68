#!/bin/bash
O=gpurun_out/c20; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python -m pytest tests/test_gpu_reid.py -q -x -k "oriented or crops_bit_exact" > $O/pytest_obb.log 2>&1
echo "pytest rc=$?" >> $O/pytest_obb.log
tail -n 12 $O/pytest_obb.log | cut -c1-300
