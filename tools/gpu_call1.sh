#!/bin/bash
# GPU-box session 1 of round 2 (run through gpurun from the repository root)
O=gpurun_out/c1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 60 tools/_build/mfma_chain_test > $O/mfma_chain.txt 2>&1
timeout 120 tools/_build/osblock_prof_phases 4096 5 > $O/osblock_phases.txt 2>&1
timeout 120 tools/_build/osblock_prof 4096 8 > $O/osblock_plain.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_compat_abi.py tests/test_gpu_long_parity.py tests/test_gpu_reid.py -q -s -m gpu --durations=15 > $O/pytest_new.log 2>&1
echo "pytest rc=$?" >> $O/pytest_new.log
timeout 600 python tools/ab_variants.py run base head_per_crop s1_handover s2_both all > $O/ab.txt 2>&1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
tail -n 5 $O/pytest_new.log; cat $O/ab.txt; cat $O/bench.json
