#!/bin/bash
O=gpurun_out/c4; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests/ -q -m gpu --durations=12 > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -n 25 $O/pytest_all.log; cat $O/smoke.log | tail -n 2
