# Round-4 profiles of bench.py (both configs): rocprofv3 kernel stats + PMC passes -> gpurun_out/r04prof/
#   gpurun --timeout 1100 -- 'bash scripts/profile_r04.sh [mlp|grid|b16|b16x6 ...]'
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
mkdir -p $O
B="python3 $R/bench.py --no-extras --no-cpu-baseline --no-alt-precision"
for cfg in ${1:-mlp grid}; do
  EXTRA=""; CFG=$cfg
  [ "$cfg" = b16 ] && EXTRA="--precision bf16x3" && CFG=mlp
  [ "$cfg" = b16x6 ] && EXTRA="--precision bf16x6" && CFG=mlp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- $B --steps 10 --warmup 3 --config $CFG $EXTRA > $O/bench_under_rocprof_$cfg.json 2>/dev/null
  cp $O/stats_$cfg/*/*kernel_stats.csv $O/kernel_stats_$cfg.csv
  cp $O/stats_$cfg/*/*kernel_trace.csv $O/kernel_trace_$cfg.csv
  rm -rf $O/stats_$cfg
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_$cfg/sq -- $B --steps 3 --warmup 2 --config $CFG $EXTRA > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_$cfg/sq2 -- $B --steps 3 --warmup 2 --config $CFG $EXTRA > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$cfg/fetch -- $B --steps 3 --warmup 2 --config $CFG $EXTRA > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_$cfg/write -- $B --steps 3 --warmup 2 --config $CFG $EXTRA > /dev/null 2>&1
  python3 $R/scripts/pmc_sum.py $O/pmc_$cfg > $O/pmc_$cfg.json
  rm -rf $O/pmc_$cfg
  echo done $cfg
done
ls -la $O
