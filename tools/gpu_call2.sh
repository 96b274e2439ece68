#!/bin/bash
O=gpurun_out/c2; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for v in pf00 pf11 pf10 pf01 pf00 pf11; do echo "== $v"; timeout 120 tools/_build/osblock_prof_$v 4096 8 | grep -E "best|weighted"; done > $O/osblock_pf.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_compat_abi.py tests/test_gpu_long_parity.py tests/test_gpu_reid.py -q -m gpu -k "binding_drives or config2 or multistream" > $O/pytest_fix.log 2>&1
echo "pytest rc=$?" >> $O/pytest_fix.log
timeout 900 python tools/ab_variants.py run base no_prefetch pf_conv1_only pf_epi_only all base > $O/ab.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-m1 --steps 60 --warmup 20 --groups 2 > $O/bench_g2.json 2> $O/bench_g2.err
tail -n 4 $O/pytest_fix.log; cat $O/osblock_pf.txt; cat $O/ab.txt; cat $O/bench_g2.json | head -c 600
