#!/bin/bash
O=gpurun_out/c23; mkdir -p $O
for v in stem1 mul24 stem1 mul24; do echo "== $v"; timeout 120 tools/_build/osblock_prof_$v 4096 10 x; timeout 120 tools/_build/osblock_prof_$v 16384 6 x; done > $O/stem_mul24_ab.txt 2>&1
cat $O/stem_mul24_ab.txt
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python -m pytest tests/test_gpu_reid.py -q -x > $O/pytest_reid.log 2>&1; echo "pytest rc=$?" >> $O/pytest_reid.log; tail -n 5 $O/pytest_reid.log | cut -c1-200
timeout 300 python bench.py --no-side-configs > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
