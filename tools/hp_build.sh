#!/bin/bash
# Build one variant of tools/hp_prof (the fp32-grade x0.25 kernels' launch-time profiler) in its own directory and print the register /
# spill / scratch table of its k_osblock_hp instantiations (from the saved ISA):
#   tools/hp_build.sh <name> [-D<switch>=<v> ...]        -> tools/_build/hp_prof_<name>  (picked up by `tools/gpu_session.sh <tag> hpab`)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
D=$R/tools/_build/t_$N; mkdir -p $D; cd $D
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$@" -I $R/boxmot_amd/csrc $R/tools/hp_prof.hip -o $D/hp_prof -save-temps=obj 2>&1 | grep -E "error|Error" | head -5
cp $D/hp_prof $R/tools/_build/hp_prof_$N
S=$(ls $D/*gfx950*.s | head -1)
echo "== $N $*   (scratch bytes, vgprs, spilled vgprs)"
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size|name):" $S | paste - - - - | grep -E "osblock_hp|stem_resize_fused_hp|head_hp" | awk '{print "   " substr($2,1,52), $4, $6, $8}'
