#!/bin/bash
# final validation of the round: full GPU suite, smoke, the default bench (with its side lines), shipped-state c3 kernel trace
O=gpurun_out/c18; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests -q -m gpu --maxfail=8 > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 400 python bench.py --streams 512 --no-cpu-baseline --no-m1 --no-side-configs > $O/bench_s512.json 2> $O/bench_s512.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c3 -o c3 -- python $R/tools/config_bench.py --config c3 --streams 8 --steps 20 --warmup 4 --check-frames 0 > $R/$O/prof_c3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $R/$O/mfma_c3 -o p -- python $R/tools/config_bench.py --config c3 --streams 8 --steps 4 --warmup 2 --check-frames 0 > $R/$O/mfma_c3.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find $O/prof_c3 -name "*.db" | head -1) > $O/r2_c3_kernel_stats.txt 2>&1
python profiles/summarize_mfma.py $(find $O/mfma_c3 -name "*.db" | head -1) > $O/r2_mfma_busy_c3.txt 2>&1
rm -rf $O/prof_c3 $O/mfma_c3
tail -n 6 $O/pytest_all.log | cut -c1-200; tail -1 $O/smoke.log; grep -v "cpu baseline frame" $O/bench.err | tail -6; cat $O/bench.json | cut -c1-2500; cat $O/bench_s512.json | cut -c1-400; head -12 $O/r2_c3_kernel_stats.txt | cut -c1-150
