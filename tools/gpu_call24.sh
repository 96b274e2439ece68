#!/bin/bash
O=gpurun_out/c24; mkdir -p $O
BOXMOT_HIP_LIB=tools/_build/libboxmot_hip_ssprof.so timeout 300 python tools/tracker_bench.py --tracker strongsort --config c5 --streams 2 --steps 20 --warmup 110 --check-frames 3 > $O/ss_prof.jsonl 2> $O/ss_prof.err
tail -c 2500 $O/ss_prof.jsonl; tail -n 3 $O/ss_prof.err
