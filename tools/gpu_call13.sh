#!/bin/bash
O=gpurun_out/c13; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
# StrongSORT soak under the device dot rule: 60 seeds x 600 frames in 6 processes of 10 seeds
for k in 0 1 2 3 4 5; do
  (timeout 900 python tools/parity_soak.py 10 600 strongsort $((100 + 10 * k)) > $O/soak_$k.log 2>&1) &
done
# meanwhile: CLIP GEMM k-tile A/B (two interleaved rounds)
for r in 1 2; do
  timeout 200 python tools/clip_bench.py --crops 256 --iters 10 >> $O/ab.txt 2>> $O/ab.err
  BOXMOT_HIP_CLIP_BK32=1 timeout 200 python tools/clip_bench.py --crops 256 --iters 10 >> $O/ab.txt 2>> $O/ab.err
done
wait
timeout 600 python tools/parity_soak.py 6 600 strongsort_blas 154 > $O/soak_blas.log 2>&1
timeout 300 python -m pytest tests/test_gpu_strongsort.py -q -m gpu -k device_dot > $O/pytest_ss.log 2>&1
echo "rc=$?" >> $O/pytest_ss.log
cat $O/ab.txt; grep -h "MISMATCH\|mismatches" $O/soak_*.log; tail -3 $O/pytest_ss.log
