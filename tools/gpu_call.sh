#!/bin/bash
# One GPU-box call = one recipe (run through gpurun; everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/).
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh <recipe> [tag] [args...]'
# recipes
#   tests [pytest -k expression]      the -m gpu suite (or a subset) with the 40 slowest durations
#   ab CFG [CFG ...]                   tools/ab_inproc.py: same-process A/B of Runtime attributes / SVDX_* knobs inside the captured step
#   ablib OLD.so NEW.so                bench.py alternately with two builds of libsvdx.so (SVDX_LIB), three rounds
#   stall                              PMC stall counters of the main GEMM kernels (two --pmc passes; tools/stall_pmc.py)
#   insitu [variants]                  tools/tune_dump.py: every (split, tile) candidate of every NT problem timed inside real sweeps
#   ring [check race time big]         tools/ring_check.py: kernel checks, race screen and isolated timings of the tile variants
#   cond                               the frozen conditioners (VAE encode + CLIP embed) under rocprofv3 --kernel-trace --stats
#   sanity                             smoke() and the quick GPU tests (what the driver runs first)
#   evidence                           tools/round_evidence.sh <tag>: bench lines of every configuration + profiler passes
recipe=${1:?recipe}; tag=${2:-call}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
O=gpurun_out/$tag; mkdir -p $O
case $recipe in
  tests)
    timeout 1400 python -m pytest tests -m gpu -q --durations=40 ${1:+-k "$1"} > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -n 50 $O/pytest.txt | cut -c1-200 ;;
  ab)
    timeout 1200 python tools/ab_inproc.py --steps 30 --reps 3 --out $O/ab.json -- "$@" > $O/ab.txt 2>&1; echo "ab rc $?"; grep -v "^\[" $O/ab.txt | tail -n 12 ;;
  ablib)
    for rep in 1 2 3; do for lib in "$1" "$2"; do
      ms=$(SVDX_LIB=$PWD/$lib timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-roofline --no-real-loop 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
      echo "[$lib] $ms" | tee -a $O/ab_lib.txt
    done; done ;;
  stall)
    PA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
    PB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
    timeout 400 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/passA -- python tools/stall_pmc.py run > $O/passA.log 2>&1; echo "pass A rc $?"
    timeout 400 rocprofv3 --pmc $PB --kernel-trace --output-format csv -d $O/passB -- python tools/stall_pmc.py run > $O/passB.log 2>&1; echo "pass B rc $?"
    python tools/stall_pmc.py report $O/passA $O/passB > $O/stall_report.txt 2> $O/stall_report.err
    find $O -name "*.db" -delete; find $O -name "*.csv" -size +20M -delete; tail -n 60 $O/stall_report.txt ;;
  insitu)
    timeout 900 python tools/tune_dump.py --rounds 2 ${1:+--only "$1"} > $O/tune_dump.txt 2>&1; echo "rc $?"; head -n 60 $O/tune_dump.txt | cut -c1-260 ;;
  ring)
    timeout 1400 python tools/ring_check.py "$@" > $O/ring_check.txt 2>&1; echo "rc $?"; grep -v "^race.*: ok\|0 failed" $O/ring_check.txt | tail -n 40 | cut -c1-260 ;;
  cond)
    timeout 300 python tools/cond_profile.py --table > $O/cond_table.txt 2>&1; tail -n 40 $O/cond_table.txt
    timeout 600 rocprofv3 --kernel-trace -d $O/cond_prof -o cond -- python tools/cond_profile.py > $O/cond_prof.log 2>&1
    f=$(find $O/cond_prof -name "*_results.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py "$f" | head -n 30 | cut -c1-200 | tee $O/cond_kernel_stats.txt
    find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete ;;
  sanity)
    timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
    timeout 300 python -m pytest tests -m gpu -q -x -k "encoders or optim or small or capi or smoke or encode_image or tiny" > $O/gpu_subset.txt 2>&1; tail -n 3 $O/gpu_subset.txt ;;
  evidence)
    timeout 2300 bash tools/round_evidence.sh $tag ;;
  *) echo "unknown recipe $recipe"; exit 2 ;;
esac
