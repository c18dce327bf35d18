# gpurun_out/r6f (tools/gpu_r6_final.sh, merged back from the GPU box) -> the tracked summaries under profiles/
O=gpurun_out/r6f; P=profiles
cp $O/bench.json $P/r06_bench.json
for f in b1024 b2048 b512 b1024_dense0_in_mlp b2048_dense0_in_mlp b512_dense0_in_mlp b4096_dense0_as_roles b65536 cfg4 cfg5 layerwise shared cfg0_episode_envs10 cfg2loop_envs50_run1 cfg2loop_envs50_run2 cfg2loop_envs50_run3 cfg2loop_env1_run1 cfg2loop_env1_run2 cfg2loop_env1_run3 cfg2loop_env1_per_transition cfg2loop_env1_b1_predicts cfg2loop_envs50_4000steps; do cp $O/bench_$f.json $P/r06_bench_$f.json; done
cp $O/kernel_stats.txt $P/r06_kernel_stats.txt; cp $O/kernel_stats_cfg4.txt $P/r06_kernel_stats_cfg4.txt; cp $O/kernel_stats_cfg5.txt $P/r06_kernel_stats_cfg5.txt
for g in 2 4 8; do cp $O/kernel_stats_share$g.txt $P/r06_kernel_stats_share$g.txt; cp $O/mfma_share$g.txt $P/r06_pmc_share$g.txt; done
cp $O/stalls.txt $P/r06_stalls.txt
cp $O/roofline.md $P/r06_roofline.md; cp $O/roofline_cfg4.md $P/r06_roofline_configs3.md; cp $O/roofline_cfg5.md $P/r06_roofline_configs4.md
for f in dropin_profile dp_host_overhead predict_latency gputests lib_sha mlpwg_phases_b512 wgrad_roles_b512 soak_split rl_loop_kernels cpu_quota ragged_phases loop_env1_threads; do cp $O/$f.txt $P/r06_$f.txt; done
cp $O/fetch_cfg4.txt $P/r06_pmc_fetch_cfg4.txt; cp $O/write_cfg4.txt $P/r06_pmc_write_cfg4.txt; cp $O/fetch_cfg5.txt $P/r06_pmc_fetch_cfg5.txt; cp $O/write_cfg5.txt $P/r06_pmc_write_cfg5.txt
cp $O/hbm_traffic.json $P/hbm_traffic.json
# the written analyses of round 6 follow the tables they refer to
printf "\n" >> $P/r06_roofline_configs3.md; cat $P/r06_notes_configs3.md >> $P/r06_roofline_configs3.md
printf "\n" >> $P/r06_roofline_configs4.md; cat $P/r06_notes_configs4.md >> $P/r06_roofline_configs4.md
