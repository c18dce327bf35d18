# per-kernel data of the per-GPU shares of configs[3] (--workload cfg4) and configs[4] (--workload cfg5)
O=gpurun_out/r3c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in cfg4 cfg5; do
  python bench.py --workload $wl --shard-of 8 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  B="python bench.py --workload $wl --shard-of 8 --no-graph --no-cpu-baseline --no-roofline --steps 20 --warmup 3 --min-seconds 0"
  rocprofv3 --kernel-trace --stats -d $O/prof_$wl -o stats -- $B > /dev/null 2> $O/prof_$wl.err
  python tools/rocpd_summary.py $(ls $O/prof_$wl/*/*.db $O/prof_$wl/*.db 2>/dev/null | head -1) > $O/kernel_stats_$wl.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$wl -o pmc -- $B > /dev/null 2> $O/fetch_$wl.err
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$wl -o pmc -- $B > /dev/null 2> $O/write_$wl.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/mfma_$wl -o pmc -- $B > /dev/null 2> $O/mfma_$wl.err
  for d in fetch write mfma; do python tools/rocpd_summary.py $(ls $O/${d}_$wl/*/*.db $O/${d}_$wl/*.db 2>/dev/null | head -1) > $O/${d}_$wl.txt 2>&1; done
  head -24 $O/kernel_stats_$wl.txt
  cut -c1-400 $O/bench_$wl.json
done
