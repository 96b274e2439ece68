#!/bin/bash
O=gpurun_out/c30; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 200 python -m pytest tests/test_gpu_deepocsort.py tests/test_gpu_configs.py tests/test_gpu_compat_abi.py -q -x > $O/pytest_docs.log 2>&1; echo "pytest rc=$?" >> $O/pytest_docs.log; tail -n 4 $O/pytest_docs.log | cut -c1-200
timeout 100 python tools/tracker_bench.py --tracker deepocsort --config c3 --streams 8 --steps 60 --warmup 40 > $O/docs_c3.jsonl 2> $O/docs_c3.err; cut -c1-220 $O/docs_c3.jsonl
timeout 100 python tools/tracker_bench.py --tracker deepocsort --config c2 --streams 16 --steps 60 --warmup 40 > $O/docs_c2.jsonl 2> $O/docs_c2.err; cut -c1-220 $O/docs_c2.jsonl
