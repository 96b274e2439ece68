#!/bin/bash
O=gpurun_out/c32; mkdir -p $O
timeout 120 tools/_build/osblock_prof_phases 4096 4 > $O/phases.txt 2>&1
cat $O/phases.txt | cut -c1-150
