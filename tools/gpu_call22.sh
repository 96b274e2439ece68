#!/bin/bash
O=gpurun_out/c22; mkdir -p $O
for v in stem1 pipe1 stem1 pipe1; do echo "== $v"; timeout 120 tools/_build/osblock_prof_$v 4096 10 x; timeout 120 tools/_build/osblock_prof_$v 16384 6 x; done > $O/stem_pipe_ab.txt 2>&1
for v in pipe1 occ6 pipe1 occ6; do echo "== $v"; timeout 120 tools/_build/osblock_prof_$v 4096 8 | grep -E "best" | grep "osblock<1"; done > $O/stage1_occ_ab.txt 2>&1
cat $O/stem_pipe_ab.txt $O/stage1_occ_ab.txt
