# round 6 measurement pass: everything profiles/r06_* is made of, on one box, with the library that ships
O=gpurun_out/r6f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sha256sum globecom2020-resourceallocationgnn_amd/libv2xgnn.so > $O/lib_sha.txt
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
if [ "$1" != "skip-tests" ]; then
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids | grep -e "every gradient" -e passed -e failed -e FAILED -e Error | tail -20 > $O/gputests.txt; cat $O/gputests.txt
fi
# PMC passes first (hbm_traffic.json must exist before bench.py reports roofline.traffic)
B="python bench.py --no-graph --no-cpu-baseline --no-roofline --no-fast-path --no-dropin --no-other-workloads --steps 40 --warmup 5 --min-seconds 0"
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o pmc -- $B > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/write -o pmc -- $B > /dev/null 2> $O/write.err
python tools/make_traffic_json.py $(db $O/fetch) $(db $O/write) > $O/hbm_traffic.json
cp $O/hbm_traffic.json profiles/hbm_traffic.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/st1 -o pmc -- $B > /dev/null 2> $O/st1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS -d $O/st2 -o pmc -- $B > /dev/null 2> $O/st2.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_BUSY_CYCLES -d $O/st3 -o pmc -- $B > /dev/null 2> $O/st3.err
python tools/stall_table.py $(db $O/st1) $(db $O/st2) $(db $O/st3) > $O/stalls.txt 2>&1
# headline + kernel stats of the same command
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 700 $O/bench.json; echo
rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python bench.py --no-cpu-baseline --no-roofline --no-dropin --no-other-workloads --min-seconds 1 > $O/prof_bench.json 2> $O/prof.err
python tools/rocpd_summary.py $(db $O/prof) > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt
# shares of the fixed global batch: Dense-0's weight gradient as roles of k_wgrad (the library's choice) against the round-5 form
Q="python bench.py --no-cpu-baseline --no-dropin --no-other-workloads --min-seconds 2"
for g in 2 4 8; do
  $Q --shard-of $g > $O/bench_b$((4096/g)).json 2>/dev/null
  V2X_MLP_WG0=1 $Q --shard-of $g > $O/bench_b$((4096/g))_dense0_in_mlp.json 2>/dev/null
done
for g in 2 4 8; do
  W="python bench.py --shard-of $g --no-graph --no-cpu-baseline --no-roofline --no-fast-path --no-dropin --no-other-workloads --steps 40 --warmup 5 --min-seconds 0"
  rocprofv3 --kernel-trace --stats -d $O/prof_share$g -o stats -- $W > /dev/null 2> $O/prof_share$g.err
  python tools/rocpd_summary.py $(db $O/prof_share$g) > $O/kernel_stats_share$g.txt 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/mfma_share$g -o pmc -- $W > /dev/null 2> $O/mfma_share$g.err
  python tools/rocpd_summary.py $(db $O/mfma_share$g) > $O/mfma_share$g.txt 2>&1
done
PHASES_BATCH=512 python tools/mlpwg_phases.py 2>&1 | grep -v amdgpu.ids > $O/mlpwg_phases_b512.txt
for R in 0 2 3; do echo "== role $R"; V2X_WG_TS_ROLE=$R PHASES_BATCH=512 python tools/wgrad_phases.py 2>&1 | grep -v amdgpu; done > $O/wgrad_roles_b512.txt
$Q --share-weights > $O/bench_shared.json 2>/dev/null
$Q --batch 65536 --steps 20 --warmup 3 > $O/bench_b65536.json 2>/dev/null
V2X_FUSED=0 V2X_MLP_WG=0 $Q > $O/bench_layerwise.json 2>/dev/null
V2X_MLP_WG0=0 $Q > $O/bench_b4096_dense0_as_roles.json 2>/dev/null
for wl in cfg4 cfg5; do
  $Q --workload $wl --shard-of 8 > $O/bench_$wl.json 2> /dev/null
  W="python bench.py --workload $wl --shard-of 8 --no-graph --no-cpu-baseline --no-roofline --no-fast-path --steps 20 --warmup 3 --min-seconds 0"
  rocprofv3 --kernel-trace --stats -d $O/prof_$wl -o stats -- $W > /dev/null 2> $O/prof_$wl.err
  python tools/rocpd_summary.py $(db $O/prof_$wl) > $O/kernel_stats_$wl.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$wl -o pmc -- $W > /dev/null 2> $O/fetch_$wl.err
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$wl -o pmc -- $W > /dev/null 2> $O/write_$wl.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/mfma_$wl -o pmc -- $W > /dev/null 2> $O/mfma_$wl.err
  for d in fetch write mfma; do python tools/rocpd_summary.py $(db $O/${d}_$wl) > $O/${d}_$wl.txt 2>&1; done
  python tools/roofline_table2.py $O/bench_$wl.json $O/fetch_$wl.txt $O/write_$wl.txt $O/mfma_$wl.txt > $O/roofline_$wl.md 2>&1
done
python tools/ragged_phases.py 2>&1 | grep -v amdgpu.ids > $O/ragged_phases.txt
bash tools/rl_loop_kernels.sh 2>&1 | grep -v -e amdgpu.ids -e rocprofv3 | cut -c1-200 > $O/rl_loop_kernels.txt
cat /sys/fs/cgroup/cpu.max > $O/cpu_quota.txt 2>/dev/null; nproc >> $O/cpu_quota.txt
python bench.py --workload cfg0 --envs 10 > $O/bench_cfg0_episode_envs10.json 2>/dev/null
for i in 1 2 3; do python bench.py --workload cfg2loop --envs 50 --episodes 5 > $O/bench_cfg2loop_envs50_run$i.json 2>/dev/null; done
for i in 1 2 3; do python bench.py --workload cfg2loop --envs 1 --episodes 2 > $O/bench_cfg2loop_env1_run$i.json 2>/dev/null; done
V2X_RL_NATIVE_ROLLOUT=0 python bench.py --workload cfg2loop --envs 1 --episodes 2 > $O/bench_cfg2loop_env1_per_transition.json 2>/dev/null
V2X_RL_ROLLOUT_BATCH_PREDICT=0 python bench.py --workload cfg2loop --envs 1 --episodes 2 > $O/bench_cfg2loop_env1_b1_predicts.json 2>/dev/null
for T in 3 4 6 8 10 12; do echo "V2X_SIM_THREADS=$T: $(V2X_SIM_THREADS=$T python bench.py --workload cfg2loop --envs 1 --episodes 2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["split"])')"; done > $O/loop_env1_threads.txt
python bench.py --workload cfg2loop --envs 50 --episodes 200 > $O/bench_cfg2loop_envs50_4000steps.json 2>/dev/null
python tools/predict_latency.py 2>&1 | grep -v amdgpu.ids > $O/predict_latency.txt
python tools/dropin_profile.py 2>&1 | grep -v amdgpu.ids > $O/dropin_profile.txt
python tools/dp_host_overhead.py 2>&1 | grep -v -e amdgpu.ids -e "version" -e Hostname -e Librccl -e socket > $O/dp_host_overhead.txt
DP_BATCH=512 python tools/dp_host_overhead.py 2>&1 | grep -v -e amdgpu.ids -e "version" -e Hostname -e Librccl -e socket > $O/dp_host_overhead_b512.txt
SOAK_STEPS=30000 python tools/soak_split.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/soak_split.txt
python tools/roofline_table.py $O/bench.json $O/hbm_traffic.json > $O/roofline.md 2>&1
# only the summaries travel back (gpurun merges <= 64 MiB): drop the raw rocprofv3 databases
rm -rf $O/fetch $O/write $O/st1 $O/st2 $O/st3 $O/prof $O/prof_cfg4 $O/prof_cfg5 $O/fetch_cfg4 $O/write_cfg4 $O/mfma_cfg4 $O/fetch_cfg5 $O/write_cfg5 $O/mfma_cfg5 $O/prof_share2 $O/prof_share4 $O/prof_share8 $O/mfma_share2 $O/mfma_share4 $O/mfma_share8
for f in $O/bench_*.json $O/bench.json; do echo "$f: $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d.get('ms_per_step'), d.get('value'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'))")"; done
