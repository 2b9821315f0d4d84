#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace statistics and the PMC passes over bench.py, one rocprofv3
# run per pass (counters are not split over passes), each under its own time limit; stops at the first failure.
#   tools/collect_profiles.sh gpurun_out/r02x
out=$1
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o b -- $B > "$out/bench_stats.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d "$out/pmc_fetch" -o f -- $B > "$out/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$out/pmc_write" -o w -- $B > "$out/pmc_write.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d "$out/pmc_sq" -o s -- $B > "$out/pmc_sq.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d "$out/pmc_sq2" -o s -- $B > "$out/pmc_sq2.log" 2>&1 || exit 1
echo collected
