#!/bin/bash
O=gpurun_out/c3; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_strongsort.py tests/test_gpu_long_parity.py tests/test_gpu_configs.py -q -m gpu -k "strongsort or config5" > $O/pytest_ss.log 2>&1
echo "pytest rc=$?" >> $O/pytest_ss.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python $R/tools/tracker_bench.py --tracker strongsort --config c5 --streams 2 --warmup 110 --steps 20 --check-frames 0 > $R/$O/tb_c5_mfma.json 2> $R/$O/tb_c5_mfma.err
BOXMOT_HIP_SS_BANK=valu timeout 300 python $R/tools/tracker_bench.py --tracker strongsort --config c5 --streams 2 --warmup 110 --steps 20 --check-frames 0 > $R/$O/tb_c5_valu.json 2> $R/$O/tb_c5_valu.err
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c5 -o ss -- python $R/tools/tracker_bench.py --tracker strongsort --config c5 --streams 2 --warmup 110 --steps 20 --check-frames 0 > $R/$O/prof_c5.log 2>&1
timeout 300 python $R/tools/tracker_bench.py --tracker deepocsort --config c3 --streams 8 --warmup 40 --steps 60 --check-frames 0 > $R/$O/tb_c3.json 2> $R/$O/tb_c3.err
timeout 300 python $R/tools/tracker_bench.py --tracker all --config c2 --streams 16 --warmup 40 --steps 60 --check-frames 4 > $R/$O/tb_c2.json 2> $R/$O/tb_c2.err
cd $R
python profiles/summarize_rocpd.py $(ls $O/prof_c5/*/*.db 2>/dev/null | head -1) > $O/prof_c5_kernels.txt 2>&1
tail -n 3 $O/pytest_ss.log; cat $O/tb_c5_mfma.json $O/tb_c5_valu.json $O/tb_c3.json $O/tb_c2.json; head -8 $O/prof_c5_kernels.txt
