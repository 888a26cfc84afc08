#!/bin/bash
# SQ / GRBM counters per kernel of the UNet forward (GPU box, through gpurun from the repo root):
#   tools/profile_sq.sh r03   ->  gpurun_out/prof_r03_sq/{a,b}  (two separate --pmc passes, kernel-trace not combined with them)
# then tools/summarize_sq.py turns them into profiles/r03_pmc_sq_per_kernel.json.
# The program itself follows `--` (no env / bash -c hop: the profiler's preload has initialised the GPU).
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    -d $OUT/a -- python3 $R/tools/run_unet.py --iters 2 --rounds 1 > $OUT/a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM \
    -d $OUT/b -- python3 $R/tools/run_unet.py --iters 2 --rounds 1 > $OUT/b.log 2>&1 || exit 1
python3 $R/tools/summarize_sq.py $OUT $R/gpurun_out/${TAG}_pmc_sq_per_kernel.json
rm -rf $OUT/a $OUT/b
