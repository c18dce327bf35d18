# gpurun_out/r5f (tools/gpu_r5_final.sh, merged back from the GPU box) -> the tracked summaries under profiles/
O=gpurun_out/r5f; P=profiles
cp $O/bench.json $P/r05_bench.json
for f in b1024 b2048 b512 b1024_wholetiles b2048_wholetiles b512_wholetiles b65536 cfg4 cfg5 layerwise shared cfg0_episode_envs10 cfg2loop_envs50_run1 cfg2loop_envs50_run2 cfg2loop_envs50_run3 cfg2loop_envs50_4000steps cfg2loop_envs50_4000steps_numpy_sampler; do cp $O/bench_$f.json $P/r05_bench_$f.json; done
cp $O/kernel_stats.txt $P/r05_kernel_stats.txt; cp $O/kernel_stats_cfg4.txt $P/r05_kernel_stats_cfg4.txt; cp $O/kernel_stats_cfg5.txt $P/r05_kernel_stats_cfg5.txt
cp $O/kernel_stats_share4.txt $P/r05_kernel_stats_share4.txt; cp $O/kernel_stats_share8.txt $P/r05_kernel_stats_share8.txt
cp $O/mfma_share4.txt $P/r05_pmc_share4.txt; cp $O/mfma_share8.txt $P/r05_pmc_share8.txt
cp $O/stalls.txt $P/r05_stalls.txt
cp $O/roofline.md $P/r05_roofline.md; cp $O/roofline_cfg4.md $P/r05_roofline_configs3.md; cp $O/roofline_cfg5.md $P/r05_roofline_configs4.md
for f in dropin_profile dp_host_overhead dp_host_overhead_b512 predict_latency gputests lib_sha rl_sections split_phases_b512 split_phases_b1024 mlpwg_phases_b512 degree_sweep wide_tail_ab ragged_small_ab soak_split rl_sections_arrays_no_lookahead rl_loop_kernels cpu_quota; do cp $O/$f.txt $P/r05_$f.txt; done
cp $O/fetch_cfg4.txt $P/r05_pmc_fetch_cfg4.txt; cp $O/write_cfg4.txt $P/r05_pmc_write_cfg4.txt; cp $O/fetch_cfg5.txt $P/r05_pmc_fetch_cfg5.txt; cp $O/write_cfg5.txt $P/r05_pmc_write_cfg5.txt
cp $O/hbm_traffic.json $P/hbm_traffic.json
