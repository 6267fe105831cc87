set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof
mkdir -p $O
MSDF_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 $R/bench.py --no-cpu-baseline > $O/bench_forcedist.json 2> $O/bench_forcedist.err || echo FORCE_DIST_FAILED
for cfg in mlp grid; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --config $cfg > $O/bench_under_rocprof_$cfg.json 2>/dev/null
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_${cfg}/sq -- python3 $R/bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline --config $cfg > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${cfg}/fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline --config $cfg > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${cfg}/write -- python3 $R/bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline --config $cfg > /dev/null 2>&1
  python3 $R/scripts/pmc_sum.py $O/pmc_${cfg} > $O/pmc_$cfg.json
  cp $O/stats_$cfg/*/*kernel_stats.csv $O/kernel_stats_$cfg.csv
  rm -rf $O/stats_$cfg $O/pmc_$cfg
  echo done $cfg
done
ls -la $O
