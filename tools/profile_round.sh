#!/bin/bash
# One profiling pass on the GPU box (run through gpurun): rocprofv3 kernel stats + PMC passes of bench.py in exact fp32 and
# in the bf16-storage configuration, post-processed into profiles/<round>_*.  usage: tools/profile_round.sh r02
set -o pipefail
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
mkdir -p $OUT profiles
for D in f32 bf16; do
  SUF=$([ $D = f32 ] && echo "" || echo "_$D")
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$D -o s -- python bench.py --steps 5 --warmup 2 \
      --no-cpu-baseline --no-alt-modes --no-dba --no-extra-legs --single-stream --dtype $D > $OUT/stats_$D.log 2>&1 || exit 1
  cp $OUT/stats_$D/s_kernel_stats.csv profiles/${R}_kernel_stats_bench_bs12$SUF.csv
  python tools/timeline_gaps.py $OUT/stats_$D/s_kernel_trace.csv > $OUT/gaps_$D.txt 2>&1 || true
  rm -f $OUT/stats_$D/s_kernel_trace.csv
  for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    T=$(echo $C | cut -d' ' -f1)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${D}_$T -o p -- python bench.py --steps 1 --warmup 1 \
        --no-cpu-baseline --no-alt-modes --no-dba --no-extra-legs --single-stream --dtype $D > $OUT/pmc_${D}_$T.log 2>&1 || exit 1
  done
  python tools/pmc_traffic.py $OUT/pmc_${D}_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_${D}_WRITE_SIZE/p_counter_collection.csv \
      profiles/${R}_pmc_traffic$SUF.json || exit 1
  python tools/pmc_mfma_busy.py $OUT/pmc_${D}_SQ_VALU_MFMA_BUSY_CYCLES/p_counter_collection.csv profiles/${R}_pmc_mfma_busy$SUF.json || exit 1
  rm -rf $OUT/pmc_${D}_*/p_counter_collection.csv $OUT/pmc_${D}_*/p_kernel_trace.csv
done
cp profiles/${R}_* gpurun_out/ 2>/dev/null
echo profile_round done
