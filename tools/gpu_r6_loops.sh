# round 6: the DQN-loop files of the measurement pass alone (libv2xsim.so / host code changed, libv2xgnn.so did not)
O=gpurun_out/r6f; mkdir -p $O
for i in 1 2 3; do python bench.py --workload cfg2loop --envs 50 --episodes 5 > $O/bench_cfg2loop_envs50_run$i.json 2>/dev/null; done
for i in 1 2 3; do python bench.py --workload cfg2loop --envs 1 --episodes 2 > $O/bench_cfg2loop_env1_run$i.json 2>/dev/null; done
V2X_RL_NATIVE_ROLLOUT=0 python bench.py --workload cfg2loop --envs 1 --episodes 2 > $O/bench_cfg2loop_env1_per_transition.json 2>/dev/null
V2X_RL_ROLLOUT_BATCH_PREDICT=0 python bench.py --workload cfg2loop --envs 1 --episodes 2 > $O/bench_cfg2loop_env1_b1_predicts.json 2>/dev/null
for T in 3 4 6 8 10 12; do echo "V2X_SIM_THREADS=$T: $(V2X_SIM_THREADS=$T python bench.py --workload cfg2loop --envs 1 --episodes 2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["split"])')"; done > $O/loop_env1_threads.txt
python bench.py --workload cfg2loop --envs 50 --episodes 200 > $O/bench_cfg2loop_envs50_4000steps.json 2>/dev/null
python bench.py --workload cfg0 --envs 10 > $O/bench_cfg0_episode_envs10.json 2>/dev/null
bash tools/rl_loop_kernels.sh 2>&1 | grep -v -e amdgpu.ids -e rocprofv3 | cut -c1-200 > $O/rl_loop_kernels.txt
python bench.py > $O/bench.json 2> $O/bench.err
for f in $O/bench_cfg*.json $O/bench.json; do echo "$f: $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d.get('ms_per_step'), d.get('value'))")"; done; cat $O/loop_env1_threads.txt | cut -c1-40
