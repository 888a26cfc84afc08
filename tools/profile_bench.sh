#!/bin/bash
# Round profile of the bench command on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1`   -> gpurun_out/prof_<tag>/stats
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of a short run -> gpurun_out/prof_<tag>/pmc_*
# then tools/rocpd_kernel_stats.py / tools/summarize_pmc.py turn them into the files committed under profiles/.
# The program itself follows `--` (no env / bash -c hop: the profiler's preload has initialised the GPU).
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-large \
    > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.log || exit 1
DB=$(ls $OUT/stats/*/*_results.db $OUT/stats/*_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python3 $R/tools/rocpd_kernel_stats.py $DB $OUT/bench_kernel_stats.csv; else
  CSV=$(ls $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$CSV" ] && cp $CSV $OUT/bench_kernel_stats.csv; fi
SHORT="--steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-roofline --no-large"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_fetch.log || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_write.log || exit 1
python3 $R/tools/summarize_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_hbm_traffic_per_launch.json
ls -la $OUT
