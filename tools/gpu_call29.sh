#!/bin/bash
# final validation of the round: full GPU suite, default bench, StrongSORT phase clocks, kernel trace + traffic counters of the bench kernels
O=gpurun_out/c29; mkdir -p $O
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests -q -m gpu --maxfail=8 > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/pytest_all.log
tail -n 6 $O/pytest_all.log | cut -c1-200
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; echo
BOXMOT_HIP_LIB=tools/_build/libboxmot_hip_ssprof.so timeout 200 python tools/tracker_bench.py --tracker strongsort --config c5 --streams 2 --steps 20 --warmup 110 --check-frames 3 > $O/ss_prof.jsonl 2> $O/ss_prof.err
tail -c 700 $O/ss_prof.jsonl
timeout 200 python tools/tracker_bench.py --tracker strongsort --config c5 --streams 2 --steps 20 --warmup 110 --check-frames 3 > $O/ss_c5.jsonl 2> $O/ss_c5.err; cut -c1-160 $O/ss_c5.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o bench -- python $R/bench.py --no-cpu-baseline --no-side-configs > $R/$O/bench_kt.json 2> $R/$O/bench_kt.err
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/$O/fetch -o p -- python $R/tools/reid_microbench.py 4096 1 2 > $R/$O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $R/$O/write -o p -- python $R/tools/reid_microbench.py 4096 1 2 > $R/$O/write.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python profiles/summarize_rocpd.py $(db kt) > $O/kernel_stats.txt 2>&1
python profiles/summarize_pmc.py $(db fetch) $(db write) 4096 > $O/pmc_traffic.txt 2>&1
rm -rf $O/kt $O/fetch $O/write
head -14 $O/kernel_stats.txt | cut -c1-150; tail -3 $O/pmc_traffic.txt
