#!/bin/bash
O=gpurun_out/c21; mkdir -p $O
for v in stem0 stem1 stem0 stem1; do echo "== $v"; timeout 120 tools/_build/osblock_prof_$v 4096 10 x; timeout 120 tools/_build/osblock_prof_$v 16384 6 x; done > $O/stem_ab.txt 2>&1
cat $O/stem_ab.txt
