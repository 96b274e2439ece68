#!/bin/bash
O=gpurun_out/c17; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 400 python -m pytest tests/test_gpu_compat_abi.py tests/test_gpu_bytetrack.py tests/test_gpu_ingest.py tests/test_gpu_botsort.py -q -m gpu > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log | cut -c1-220
