#!/bin/bash
O=gpurun_out/c19; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests -q -m gpu --maxfail=8 > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -n 8 $O/pytest_all.log | cut -c1-200
