#!/bin/bash
O=gpurun_out/c7; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_clipreid.py tests/test_gpu_long_parity.py -q -m gpu -x -s -k "clip or x1_0" > $O/pytest_new.log 2>&1
echo "pytest rc=$?" >> $O/pytest_new.log
timeout 300 python tools/clip_bench.py --crops 256 --iters 10 > $O/clip_bench.json 2> $O/clip_bench.err
timeout 600 python tools/config_bench.py --config c3 --streams 8 > $O/c3.json 2> $O/c3.err
timeout 600 python tools/config_bench.py --config c5 --streams 2 --steps 12 --warmup 4 > $O/c5.json 2> $O/c5.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c3 -o c3 -- python $R/tools/config_bench.py --config c3 --streams 8 --steps 20 --warmup 4 --check-frames 0 > $R/$O/prof_c3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_clip -o clip -- python $R/tools/clip_bench.py --crops 256 --iters 10 > $R/$O/prof_clip.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find $O/prof_c3 -name "*.db" | head -1) > $O/prof_c3_kernels.txt 2>&1
python profiles/summarize_rocpd.py $(find $O/prof_clip -name "*.db" | head -1) > $O/prof_clip_kernels.txt 2>&1
tail -n 8 $O/pytest_new.log; cat $O/clip_bench.json; tail -2 $O/clip_bench.err; cat $O/c3.json; tail -3 $O/c3.err; cat $O/c5.json; tail -3 $O/c5.err; head -22 $O/prof_c3_kernels.txt; head -12 $O/prof_clip_kernels.txt
