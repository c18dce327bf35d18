# gpurun_out/r4f (tools/gpu_r4_final.sh, merged back from the GPU box) -> the tracked summaries under profiles/
O=gpurun_out/r4f; P=profiles
cp $O/bench.json $P/r04_bench.json
for f in b1024 b2048 b512 b65536 cfg4 cfg5 cfg5_intervalplan layerwise shared cfg0_episode_envs10 cfg2loop_envs50_run1 cfg2loop_envs50_run2 cfg2loop_envs50_run3; do cp $O/bench_$f.json $P/r04_bench_$f.json; done
cp $O/kernel_stats.txt $P/r04_kernel_stats.txt; cp $O/kernel_stats_cfg4.txt $P/r04_kernel_stats_cfg4.txt; cp $O/kernel_stats_cfg5.txt $P/r04_kernel_stats_cfg5.txt
cp $O/stalls.txt $P/r04_stalls.txt; cat $O/lds_form0.txt $O/lds_form1.txt > $P/r04_lds_forms.txt
cp $O/roofline.md $P/r04_roofline.md; cp $O/roofline_cfg4.md $P/r04_roofline_configs3.md; cp $O/roofline_cfg5.md $P/r04_roofline_configs4.md
for f in dropin_profile dp_host_overhead dp_host_overhead_wide predict_latency gputests lib_sha ragged_phases rl_sections sim_threads; do cp $O/$f.txt $P/r04_$f.txt; done
cp $O/fetch_cfg4.txt $P/r04_pmc_fetch_cfg4.txt; cp $O/write_cfg4.txt $P/r04_pmc_write_cfg4.txt; cp $O/fetch_cfg5.txt $P/r04_pmc_fetch_cfg5.txt; cp $O/write_cfg5.txt $P/r04_pmc_write_cfg5.txt
cp $O/hbm_traffic.json $P/hbm_traffic.json
