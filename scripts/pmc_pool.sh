#!/bin/bash
# PMC passes over the pooling kernels (separate passes: counter slots, MI355X_MICROARCH.md "rocprofv3 PMC slots")
# usage: scripts/pmc_pool.sh <variant-word> <out-dir-under-gpurun_out>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-mfma}
O=$R/gpurun_out/${2:-pmc_pool}
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $O/sq -o sq --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS -d $O/sq2 -o sq2 --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/sq2.log 2>&1 || true
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/tcc -o tcc --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/tcc2 -o tcc2 --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/tcc2.log 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/grbm -o grbm --output-format csv -- python3 $R/scripts/bench_pool.py "$V" > $O/grbm.log 2>&1 || true
echo done
