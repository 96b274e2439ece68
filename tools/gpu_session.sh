#!/bin/bash
# One GPU-box session, built from named steps (run through gpurun from the repo root):
#
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r3a build tests bench trace traffic'
#
#   tools/gpu_session.sh <tag> <step> [<step> ...]          outputs under gpurun_out/<tag>/
#
# steps
#   build      compile the tree's sources ON THIS BOX (BOXMOT_FORCE_BUILD=1), so the binary that runs is not a prebuilt one
#   tests      python -m pytest tests -m gpu                                  -> pytest.log
#   fast       python -m pytest tests -m "gpu and fast" (tests/conftest.py FAST_GPU: about a minute) -> pytest_fast.log
#   costs      tests/test_gpu_cost_values.py (association cost values vs the reference's matrices) -> pytest_costs.log
#   reid       the ReID GPU tests only                                        -> pytest_reid.log
#   bench      default bench.py                                               -> bench.json / bench.err
#   benchq     bench.py without CPU baseline / side lines (quick)             -> benchq.json
#   trace      rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-side-configs --no-m1` -> kernel_stats.txt
#   traffic    FETCH_SIZE and WRITE_SIZE, separate --pmc passes, on tools/reid_microbench.py 4096 crops (mode $REID_MODE)  -> pmc_traffic.txt
#   mfma       SQ_VALU_MFMA_BUSY_CYCLES pass on the same microbenchmark      -> mfma_busy.txt
#   sq         wave-time counters (SQ_WAVE_CYCLES / WAIT / ACTIVE / INSTS, four --pmc passes) on the same microbenchmark -> hp_sq_counters.txt
#   c3trace    kernel trace + FETCH_SIZE / WRITE_SIZE passes of tools/_build/wide_hp_prof 1024 (configuration 3's ReID pass) -> c3_kernel_stats.txt, c3_pmc_traffic.txt
#   c5trace    kernel trace of tools/clip_bench.py --crops 512 (configuration 5's ReID pass) -> clip_kernel_stats.txt
#   c3 / c5    tools/config_bench.py for configurations 3 / 5                 -> config_bench.jsonl
#   soak       tools/parity_soak.py (all trackers, short)                     -> soak.log
#   groups     tools/config_bench.py for configurations 3 and 5 with 1 and 2 stream groups (no id gate)     -> config_groups.jsonl
#   warps      the device-step warp, StrongSORT and ReID GPU tests                -> pytest_warps.log
#   ingest     tests/test_gpu_ingest.py                                        -> pytest_ingest.log
#   hpab       every tools/_build/hp_prof_* binary (variants of the fp32-grade kernels built with -D switches) -> hp_ab.txt
#   obb        the oriented-detection GPU tests + the reference-named ABI file, then tools/obb_step_time.py         -> pytest_obb.log, obb_step_time.txt
# Counters are collected in their own --pmc passes, never together with a trace (profiles/README.md).  Summaries a round wants
# judged are copied from gpurun_out/<tag>/ into profiles/ by hand.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
MODE=${REID_MODE:-2}
db() { find $O/$1 -name "*.db" | head -1; }
# first line of every profile summary: the source hash of the library it was taken with (bench.py reports a profile's figure only while
# that hash equals the shipped library's; a reader can tell a stale summary from a current one)
stamp() { python -c "import json; print('# source_hash: ' + json.load(open('boxmot_amd/libboxmot_hip.so.buildinfo'))['source_hash'] + '   (boxmot_amd/libboxmot_hip.so.buildinfo of the library this profile was taken with)')"; }
cd $R
for step in "$@"; do
  echo "=== $step"
  case $step in
    build)   BOXMOT_FORCE_BUILD=1 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -n 2 $O/build.log ;;
    tests)   timeout 1200 python -m pytest tests -q -m gpu --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -n 8 $O/pytest.log | cut -c1-220 ;;
    fast)    timeout 400 python -m pytest tests -q -m "gpu and fast" --durations=8 > $O/pytest_fast.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fast.log; tail -n 16 $O/pytest_fast.log | cut -c1-220 ;;
    costs)   timeout 400 python -m pytest tests/test_gpu_cost_values.py -q -s > $O/pytest_costs.log 2>&1; echo "pytest rc=$?" >> $O/pytest_costs.log; grep -E "max \||passed|failed|rc=|Error" $O/pytest_costs.log | cut -c1-220 ;;
    reid)    timeout 600 python -m pytest tests/test_gpu_reid.py -q -s --maxfail=10 > $O/pytest_reid.log 2>&1; echo "pytest rc=$?" >> $O/pytest_reid.log; grep -E "calibrated|passed|failed|rc=" $O/pytest_reid.log | cut -c1-220 ;;
    bench)   timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; echo ;;
    benchq)  timeout 400 python bench.py --no-cpu-baseline --no-side-configs --no-m1 --reid-mode $MODE > $O/benchq_m$MODE.json 2> $O/benchq.err; tail -c 900 $O/benchq_m$MODE.json; echo ;;
    trace)   (cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt -o bench -- python $R/bench.py --no-cpu-baseline --no-side-configs --no-m1 --reid-mode $MODE > $O/bench_kt.json 2> $O/bench_kt.err)
             stamp > $O/kernel_stats_m$MODE.txt
             python profiles/summarize_rocpd.py $(db kt) >> $O/kernel_stats_m$MODE.txt 2>&1; rm -rf $O/kt; head -n 17 $O/kernel_stats_m$MODE.txt | cut -c1-170 ;;
    traffic) (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p -- python $R/tools/reid_microbench.py 4096 $MODE 2 > $O/fetch.log 2>&1
              timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p -- python $R/tools/reid_microbench.py 4096 $MODE 2 > $O/write.log 2>&1)
             # (the library's source hash goes on the first line: bench.py reports this file's figure only while the hashes agree)
             stamp > $O/pmc_traffic_m$MODE.txt
             python profiles/summarize_pmc.py $(db fetch) $(db write) 4096 >> $O/pmc_traffic_m$MODE.txt 2>&1; rm -rf $O/fetch $O/write; tail -n 14 $O/pmc_traffic_m$MODE.txt ;;
    mfma)    (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -o p -- python $R/tools/reid_microbench.py 4096 $MODE 2 > $O/mfma.log 2>&1)
             stamp > $O/mfma_busy_m$MODE.txt
             python profiles/summarize_mfma.py $(db mfma) >> $O/mfma_busy_m$MODE.txt 2>&1; rm -rf $O/mfma; cat $O/mfma_busy_m$MODE.txt ;;
    sq)      # what a wave does with its time: four --pmc passes (no trace) on the ReID microbenchmark -> hp_sq_counters.txt
             stamp > $O/hp_sq_counters_m$MODE.txt
             i=0
             for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
                        "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_SALU" \
                        "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" \
                        "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
               i=$((i+1))
               (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $set -d $O/sq$i -o p -- python $R/tools/reid_microbench.py 4096 $MODE 2 > $O/sq$i.log 2>&1)
               echo "## pass $i: $set" >> $O/hp_sq_counters_m$MODE.txt
               python profiles/summarize_sq.py $(db sq$i) >> $O/hp_sq_counters_m$MODE.txt 2>&1; rm -rf $O/sq$i
             done
             grep -c SQ_ $O/hp_sq_counters_m$MODE.txt ;;
    c3trace) # configuration 3's ReID pass (fp32-grade osnet_x1_0 family): per-kernel table + HBM traffic, on tools/_build/wide_hp_prof (built from this tree)
             (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/c3kt -o p -- $R/tools/_build/wide_hp_prof 1024 4 > $O/c3_wide_hp_prof.txt 2>&1)
             stamp > $O/c3_kernel_stats.txt; grep -E "forward|checksum" $O/c3_wide_hp_prof.txt | sed 's/^/# /' >> $O/c3_kernel_stats.txt
             python profiles/summarize_rocpd.py $(db c3kt) >> $O/c3_kernel_stats.txt 2>&1; rm -rf $O/c3kt
             (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/c3f -o p -- $R/tools/_build/wide_hp_prof 1024 2 > $O/c3f.log 2>&1
              timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/c3w -o p -- $R/tools/_build/wide_hp_prof 1024 2 > $O/c3w.log 2>&1)
             stamp > $O/c3_pmc_traffic.txt
             python profiles/summarize_pmc.py $(db c3f) $(db c3w) 1024 >> $O/c3_pmc_traffic.txt 2>&1; rm -rf $O/c3f $O/c3w
             head -n 12 $O/c3_kernel_stats.txt | cut -c1-150; tail -n 3 $O/c3_pmc_traffic.txt ;;
    c5trace) # configuration 5's ReID pass (CLIP-ReID ViT-B/16, 512 crops): per-kernel table
             (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/c5kt -o p -- python $R/tools/clip_bench.py --crops 512 --iters 8 > $O/clip_bench.txt 2>&1)
             stamp > $O/clip_kernel_stats.txt; tail -n 1 $O/clip_bench.txt | sed 's/^/# /' >> $O/clip_kernel_stats.txt
             python profiles/summarize_rocpd.py $(db c5kt) >> $O/clip_kernel_stats.txt 2>&1; rm -rf $O/c5kt; head -n 10 $O/clip_kernel_stats.txt | cut -c1-150 ;;
    c3)      timeout 600 python tools/config_bench.py --config c3 --reid-mode ${C3_MODE:-2} >> $O/config_bench.jsonl 2> $O/c3.err; tail -n 1 $O/config_bench.jsonl | cut -c1-900 ;;
    c5)      timeout 900 python tools/config_bench.py --config c5 >> $O/config_bench.jsonl 2> $O/c5.err; tail -n 1 $O/config_bench.jsonl | cut -c1-900 ;;
    soak)    timeout 900 python tools/parity_soak.py 10 200 > $O/soak.log 2>&1; tail -n 12 $O/soak.log ;;
    groups)  timeout 800 python -c "
import sys, json; sys.path.insert(0, 'tools'); import config_bench as cb
for cfg, kw in (('c3', dict(steps=16, warmup=6)), ('c5', dict(steps=8, warmup=104))):
    for g in (1, 2):
        print(json.dumps(cb.run(cfg, check_frames=0, groups=g, **kw)), flush=True)
" >> $O/config_groups.jsonl 2> $O/groups.err; cut -c1-420 $O/config_groups.jsonl ;;
    warps)   timeout 600 python -m pytest tests/test_gpu_device_step_warps.py tests/test_gpu_strongsort.py tests/test_gpu_reid.py -q > $O/pytest_warps.log 2>&1; tail -n 3 $O/pytest_warps.log ;;
    ingest)  timeout 400 python -m pytest tests/test_gpu_ingest.py -q > $O/pytest_ingest.log 2>&1; tail -n 3 $O/pytest_ingest.log ;;
    obb)     timeout 400 python -m pytest tests/test_gpu_obb.py tests/test_gpu_compat_abi.py -q > $O/pytest_obb.log 2>&1; tail -n 3 $O/pytest_obb.log
             timeout 120 python tools/obb_step_time.py 150 > $O/obb_step_time.txt 2>&1; grep -v amdgpu.ids $O/obb_step_time.txt | tail -n 5 ;;
    hpab)    for b in tools/_build/hp_prof_*; do echo "## $b" >> $O/hp_ab.txt; timeout 120 $b 4096 5 >> $O/hp_ab.txt 2>&1; done; grep -c best $O/hp_ab.txt ;;
    *)       echo "unknown step $step" ;;
  esac
done
