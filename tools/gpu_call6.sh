#!/bin/bash
O=gpurun_out/c6; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_clipreid.py -q -m gpu -x -s > $O/pytest_clip.log 2>&1
echo "pytest rc=$?" >> $O/pytest_clip.log
timeout 300 python tools/clip_bench.py --crops 256 --iters 10 > $O/clip_bench.json 2> $O/clip_bench.err
timeout 300 python tools/clip_bench.py --crops 1024 --iters 5 >> $O/clip_bench.json 2>> $O/clip_bench.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_clip -o clip -- python $R/tools/clip_bench.py --crops 256 --iters 10 > $R/$O/prof_clip.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(ls $O/prof_clip/*/*.db 2>/dev/null | head -1) > $O/prof_clip_kernels.txt 2>&1
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -n 12 $O/pytest_clip.log; cat $O/clip_bench.json; tail -3 $O/clip_bench.err; head -14 $O/prof_clip_kernels.txt; tail -n 4 $O/pytest_all.log
