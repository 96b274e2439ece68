#!/bin/bash
O=gpurun_out/c16; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
B4=$PWD/tools/_build/libboxmot_hip_band4.so
for r in 1 2; do
  timeout 200 python tools/config_bench.py --config c3 --streams 8 --check-frames 0 >> $O/ab.txt 2>> $O/ab.err
  BOXMOT_HIP_LIB=$B4 timeout 200 python tools/config_bench.py --config c3 --streams 8 --check-frames 0 >> $O/ab.txt 2>> $O/ab.err
done
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/c16/ab.txt')):
    d=json.loads(l); print('band4' if i%2 else 'band8', round(d['frames_per_s'],1), round(d['reid_forward_ms_per_step'],3), d.get('reid_max_abs_err_vs_fp32_oracle'))
PY
tail -2 $O/ab.err
