#!/bin/bash
# round-2 profile session: kernel trace of the default bench, HBM traffic and matrix-pipe counters (separate --pmc passes)
O=gpurun_out/r2; mkdir -p $O
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o bench -- python $R/bench.py --no-cpu-baseline > $R/$O/bench_kt.json 2> $R/$O/bench_kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$O/fetch -o p -- python $R/tools/reid_microbench.py 4096 1 2 > $R/$O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$O/write -o p -- python $R/tools/reid_microbench.py 4096 1 2 > $R/$O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $R/$O/mfma -o p -- python $R/tools/reid_microbench.py 4096 1 2 > $R/$O/mfma.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $R/$O/mfma_clip -o p -- python $R/tools/clip_bench.py --crops 256 --iters 3 > $R/$O/mfma_clip.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $R/$O/mfma_c3 -o p -- python $R/tools/config_bench.py --config c3 --streams 8 --steps 4 --warmup 2 --check-frames 0 > $R/$O/mfma_c3.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python profiles/summarize_rocpd.py $(db kt) > $O/r2_kernel_stats.txt 2>&1
python profiles/summarize_pmc.py $(db fetch) $(db write) 4096 > $O/r2_pmc_traffic.txt 2>&1
python profiles/summarize_mfma.py $(db mfma) > $O/r2_mfma_busy.txt 2>&1
python profiles/summarize_mfma.py $(db mfma_clip) > $O/r2_mfma_busy_clip.txt 2>&1
python profiles/summarize_mfma.py $(db mfma_c3) > $O/r2_mfma_busy_c3.txt 2>&1
rm -rf $O/kt $O/fetch $O/write $O/mfma $O/mfma_clip $O/mfma_c3
cat $O/bench_kt.json | head -c 600; echo; head -14 $O/r2_kernel_stats.txt; cat $O/r2_pmc_traffic.txt; cat $O/r2_mfma_busy.txt; head -8 $O/r2_mfma_busy_clip.txt; head -12 $O/r2_mfma_busy_c3.txt
