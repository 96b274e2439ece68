#!/bin/bash
O=gpurun_out/c10; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python -m pytest tests/test_gpu_ingest.py -q -m gpu -s > $O/pytest_ecc.log 2>&1
echo "pytest rc=$?" >> $O/pytest_ecc.log
timeout 1200 python -m pytest tests -q -m gpu --maxfail=6 > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python tools/config_bench.py --config c5 --streams 2 --steps 12 --warmup 4 > $O/c5.json 2> $O/c5.err
tail -n 12 $O/pytest_ecc.log; tail -n 12 $O/pytest_all.log; tail -2 $O/smoke.log; cat $O/bench.json; tail -3 $O/bench.err; cat $O/c5.json; tail -2 $O/c5.err
