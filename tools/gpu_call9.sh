#!/bin/bash
O=gpurun_out/c9; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_clipreid.py tests/test_gpu_long_parity.py -q -m gpu -x -s -k "vitb16_features or x1_0_fp16" > $O/pytest_new.log 2>&1
echo "pytest rc=$?" >> $O/pytest_new.log
timeout 300 python tools/clip_bench.py --crops 256 --iters 10 > $O/clip_bench.json 2> $O/clip_bench.err
timeout 600 python tools/config_bench.py --config c3 --streams 8 --check-frames 0 > $O/c3.json 2> $O/c3.err
timeout 300 python tools/host_api_bench.py --streams 16 > $O/host_api.json 2> $O/host_api.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c3 -o c3 -- python $R/tools/config_bench.py --config c3 --streams 8 --steps 20 --warmup 4 --check-frames 0 > $R/$O/prof_c3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_clip -o clip -- python $R/tools/clip_bench.py --crops 256 --iters 10 > $R/$O/prof_clip.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find $O/prof_c3 -name "*.db" | head -1) > $O/prof_c3_kernels.txt 2>&1
python profiles/summarize_rocpd.py $(find $O/prof_clip -name "*.db" | head -1) > $O/prof_clip_kernels.txt 2>&1
tail -n 5 $O/pytest_new.log; cat $O/clip_bench.json; tail -2 $O/clip_bench.err; cat $O/c3.json; tail -3 $O/c3.err; cat $O/host_api.json; tail -3 $O/host_api.err; head -20 $O/prof_c3_kernels.txt; head -8 $O/prof_clip_kernels.txt
