#!/bin/bash
O=gpurun_out/c25; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
BOXMOT_HIP_LIB=tools/_build/libboxmot_hip_ssprof.so timeout 300 python tools/tracker_bench.py --tracker strongsort --config c5 --streams 2 --steps 20 --warmup 110 --check-frames 3 > $O/ss_prof.jsonl 2> $O/ss_prof.err
tail -c 1500 $O/ss_prof.jsonl
timeout 300 python tools/tracker_bench.py --tracker strongsort --config c5 --streams 2 --steps 20 --warmup 110 --check-frames 3 > $O/ss_c5.jsonl 2> $O/ss_c5.err; cut -c1-200 $O/ss_c5.jsonl
timeout 300 python tools/tracker_bench.py --tracker strongsort --config c2 --streams 16 --steps 60 --warmup 40 > $O/ss_c2.jsonl 2> $O/ss_c2.err; cut -c1-200 $O/ss_c2.jsonl
timeout 600 python -m pytest tests/test_gpu_strongsort.py tests/test_gpu_configs.py -q -x > $O/pytest_ss.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ss.log; tail -n 5 $O/pytest_ss.log | cut -c1-200
