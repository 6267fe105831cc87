# rocprofv3 passes over the hash-grid embedding scatter at configs[2] size (scripts/bench_hash_scatter.py, fused entry only)
#   gpurun -- 'bash scripts/profile_hash.sh r03x'   -> gpurun_out/<tag>/pmc_scatter.json, scatter_kernel_stats.csv
set -e
TAG=${1:-r03hash}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ONLY=binned_fused
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/scripts/bench_hash_scatter.py > $O/bench_under_rocprof.json 2>/dev/null
cp $O/stats/*/*kernel_stats.csv $O/scatter_kernel_stats.csv && rm -rf $O/stats
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc/sq1 -- python3 $R/scripts/bench_hash_scatter.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/pmc/sq2 -- python3 $R/scripts/bench_hash_scatter.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/fetch -- python3 $R/scripts/bench_hash_scatter.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/write -- python3 $R/scripts/bench_hash_scatter.py > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc/tcc -- python3 $R/scripts/bench_hash_scatter.py > /dev/null 2>&1
python3 $R/scripts/pmc_sum.py $O/pmc hb2 > $O/pmc_scatter.json
python3 $R/scripts/pmc_sum.py $O/pmc hg_forward > $O/pmc_forward.json
rm -rf $O/pmc
cat $O/pmc_scatter.json
