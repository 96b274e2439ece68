#!/bin/bash
# StrongSORT configuration-5 kernel trace after the frame-step rework + smoke()
O=gpurun_out/c31; mkdir -p $O
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o ss -- python $R/tools/tracker_bench.py --tracker strongsort --config c5 --streams 2 --steps 20 --warmup 110 --check-frames 0 > $R/$O/ss_kt.jsonl 2> $R/$O/ss_kt.err
cd $R
python profiles/summarize_rocpd.py $(find $O/kt -name "*.db" | head -1) > $O/ss_c5_kernel_stats.txt 2>&1
rm -rf $O/kt
head -8 $O/ss_c5_kernel_stats.txt | cut -c1-170
