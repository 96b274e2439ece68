#!/bin/bash
O=gpurun_out/c15; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 150 python -m pytest tests/test_gpu_long_parity.py -q -m gpu -s -x -k "x1_0" > $O/pytest_x1.log 2>&1
rc=$?
echo "rc=$rc" >> $O/pytest_x1.log
if [ $rc -ne 0 ]; then tail -25 $O/pytest_x1.log | cut -c1-250; exit 0; fi
timeout 200 python tools/config_bench.py --config c3 --streams 8 > $O/c3.json 2> $O/c3.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c3 -o c3 -- python $R/tools/config_bench.py --config c3 --streams 8 --steps 20 --warmup 4 --check-frames 0 > $R/$O/prof_c3.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find $O/prof_c3 -name "*.db" | head -1) > $O/prof_c3_kernels.txt 2>&1
rm -rf $O/prof_c3
tail -4 $O/pytest_x1.log; cat $O/c3.json; tail -2 $O/c3.err; head -18 $O/prof_c3_kernels.txt
