#!/bin/bash
# Round profile of bench.py itself (GPU box): kernel stats of the default command + PMC passes for the pooling kernel at
# the bench's own voxel count.  Separate passes per counter group (MI355X_MICROARCH.md "rocprofv3 PMC slots"); --pmc is
# never combined with tracing domains other than --kernel-trace.
# usage: scripts/profile_bench.sh <tag>      -> gpurun_out/<tag>/{stats,fetch,write,tcc,sq}/...
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r02}
O=$R/gpurun_out/$T
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-train > $O/bench_under_rocprof.json 2> $O/stats.log
B="python3 $R/bench.py --steps 2 --warmup 1 --scenes 1 --streams 1 --no-cpu-baseline --no-train"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o c --output-format csv -- $B > $O/fetch.json 2> $O/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o c --output-format csv -- $B > $O/write.json 2> $O/write.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/tcc -o c --output-format csv -- $B > $O/tcc.json 2> $O/tcc.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $O/sq -o c --output-format csv -- $B > $O/sq.json 2> $O/sq.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/grbm -o c --output-format csv -- $B > $O/grbm.json 2> $O/grbm.log
python3 $R/scripts/pmc_summarize.py $O cs_pool_kernel > $O/pool_pmc_summary.json
python3 $R/scripts/pmc_summarize.py $O conv_phase > $O/conv_pmc_summary.json
python3 $R/scripts/pmc_conv_layer.py $O/conv_pmc_summary.json $O/fetch.json > $O/conv_pmc_per_layer.json
python3 $R/scripts/pmc_summarize.py $O affinity_cs_kernel affinity_block_kernel knn_ring_kernel lift_masks_views_kernel lv_sort_scores_kernel cs_fill_kernel \
        cs_count_kernel transpose_views_kernel nn_grid_query_wave_kernel embed_head_kernel classify16 scatter_mean fuse_top3 > $O/small_pmc_summary.json
# keep the summaries only: per-dispatch traces and counter dumps are tens of MiB (gpurun_out/ is capped at 64 MiB)
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*_counter_collection.csv" -delete
echo done
