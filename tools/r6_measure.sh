# round 6, measurements that need no code change: per-role HBM fetch of the configs[3] weight-gradient roles (unmerged: one launch
# per layer), the DQN loop's kernel statistics, the ragged forward's phase stamps, the data-parallel forms at the 512-graph share
O=gpurun_out/r6m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
W="python bench.py --workload cfg4 --shard-of 8 --no-graph --no-cpu-baseline --no-roofline --no-fast-path --steps 12 --warmup 3 --min-seconds 0"
V2X_WIDE_MERGE=0 V2X_WIDE_ADAM=0 rocprofv3 --pmc FETCH_SIZE -d $O/fetch_unmerged -o pmc -- $W > /dev/null 2> $O/fetch_unmerged.err
python tools/rocpd_summary.py $(db $O/fetch_unmerged) > $O/fetch_cfg4_unmerged.txt 2>&1
V2X_WIDE_MERGE=0 V2X_WIDE_ADAM=0 rocprofv3 --kernel-trace --stats -d $O/prof_unmerged -o stats -- $W > /dev/null 2> $O/prof_unmerged.err
python tools/rocpd_summary.py $(db $O/prof_unmerged) > $O/kernel_stats_cfg4_unmerged.txt 2>&1
bash tools/rl_loop_kernels.sh 2>&1 | grep -v -e amdgpu.ids -e rocprofv3 | cut -c1-200 > $O/rl_loop_kernels.txt
python tools/ragged_phases.py 2>&1 | grep -v amdgpu.ids > $O/ragged_phases.txt
DP_BATCH=512 python tools/dp_host_overhead.py 2>&1 | grep -v -e amdgpu.ids -e "version" -e Hostname -e Librccl > $O/dp_host_overhead_b512.txt
DP_BATCH=1024 python tools/dp_host_overhead.py 2>&1 | grep -v -e amdgpu.ids -e "version" -e Hostname -e Librccl > $O/dp_host_overhead_b1024.txt
rm -rf $O/fetch_unmerged $O/prof_unmerged
head -30 $O/fetch_cfg4_unmerged.txt; head -24 $O/kernel_stats_cfg4_unmerged.txt; cat $O/rl_loop_kernels.txt $O/ragged_phases.txt $O/dp_host_overhead_b512.txt $O/dp_host_overhead_b1024.txt
