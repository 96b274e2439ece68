#!/bin/bash
O=gpurun_out/c11; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
P3=$PWD/tools/_build/libboxmot_hip_pipe3.so
for r in 1 2; do
  timeout 200 python tools/clip_bench.py --crops 256 --iters 10 >> $O/ab.txt 2>> $O/ab.err
  BOXMOT_HIP_LIB=$P3 timeout 200 python tools/clip_bench.py --crops 256 --iters 10 >> $O/ab.txt 2>> $O/ab.err
done
BOXMOT_HIP_LIB=$P3 timeout 300 python -m pytest tests/test_gpu_clipreid.py -q -m gpu -x -s -k vitb16_features > $O/pytest_p3.log 2>&1
echo "rc=$?" >> $O/pytest_p3.log
cat $O/ab.txt; tail -4 $O/pytest_p3.log
