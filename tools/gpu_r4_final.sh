# round 4 measurement pass: everything profiles/r04_* is made of, on one box, with the library that ships
O=gpurun_out/r4f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sha256sum globecom2020-resourceallocationgnn_amd/libv2xgnn.so > $O/lib_sha.txt
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
if [ "$1" != "skip-tests" ]; then
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids | grep -e "every gradient" -e passed -e failed -e FAILED -e Error | tail -14 > $O/gputests.txt; cat $O/gputests.txt
fi
# PMC passes first (hbm_traffic.json must exist before bench.py reports roofline.traffic).  The headline engine runs the
# edge-index gather in its fused graph layers; V2X_FUSED_COMPL=1 = the complement fast path.
B="python bench.py --no-graph --no-cpu-baseline --no-roofline --no-fast-path --no-dropin --no-other-workloads --steps 40 --warmup 5 --min-seconds 0"
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o pmc -- $B > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/write -o pmc -- $B > /dev/null 2> $O/write.err
python tools/make_traffic_json.py $(db $O/fetch) $(db $O/write) > $O/hbm_traffic.json
cp $O/hbm_traffic.json profiles/hbm_traffic.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/st1 -o pmc -- $B > /dev/null 2> $O/st1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS -d $O/st2 -o pmc -- $B > /dev/null 2> $O/st2.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_BUSY_CYCLES -d $O/st3 -o pmc -- $B > /dev/null 2> $O/st3.err
python tools/stall_table.py $(db $O/st1) $(db $O/st2) $(db $O/st3) > $O/stalls.txt 2>&1
# LDS work of the two aggregation forms of the fused graph layers
for form in 0 1; do
  V2X_FUSED_COMPL=$form rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/lds$form -o pmc -- $B > /dev/null 2> $O/lds$form.err
  python tools/rocpd_summary.py $(db $O/lds$form) 2>&1 | grep -e "^kernel" -e k_gnn_ > $O/lds_form$form.txt
done
# headline + kernel stats of the same command
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json; echo
rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python bench.py --no-cpu-baseline --no-roofline --no-dropin --no-other-workloads --min-seconds 1 > $O/prof_bench.json 2> $O/prof.err
python tools/rocpd_summary.py $(db $O/prof) > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt
# shares of the fixed global batch, other workloads
Q="python bench.py --no-cpu-baseline --no-dropin --no-other-workloads --min-seconds 2"
for g in 2 4 8; do $Q --shard-of $g > $O/bench_b$((4096/g)).json 2>/dev/null; done
$Q --share-weights > $O/bench_shared.json 2>/dev/null
$Q --batch 65536 --steps 20 --warmup 3 > $O/bench_b65536.json 2>/dev/null
V2X_FUSED=0 V2X_MLP_WG=0 $Q > $O/bench_layerwise.json 2>/dev/null
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
V2X_RAGGED_PACKED=0 $Q --workload cfg5 --shard-of 8 > $O/bench_cfg5_intervalplan.json 2>/dev/null
python tools/ragged_phases.py 2>&1 | grep -v amdgpu.ids > $O/ragged_phases.txt
python tools/prof_rl_sections.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/rl_sections.txt
python tools/sim_threads.py 2>&1 | grep -v amdgpu.ids > $O/sim_threads.txt
python bench.py --workload cfg0 --envs 10 > $O/bench_cfg0_episode_envs10.json 2>/dev/null
for i in 1 2 3; do python bench.py --workload cfg2loop --envs 50 > $O/bench_cfg2loop_envs50_run$i.json 2>/dev/null; done
python tools/predict_latency.py 2>&1 | grep -v amdgpu.ids > $O/predict_latency.txt
python tools/dropin_profile.py 2>&1 | grep -v amdgpu.ids > $O/dropin_profile.txt
python tools/dp_host_overhead.py --wide 2>&1 | grep -v -e amdgpu.ids -e "version" -e Hostname -e Librccl > $O/dp_host_overhead_wide.txt
python tools/dp_host_overhead.py 2>&1 | grep -v -e amdgpu.ids -e "version" -e Hostname -e Librccl > $O/dp_host_overhead.txt
python tools/roofline_table.py $O/bench.json $O/hbm_traffic.json > $O/roofline.md 2>&1
# only the summaries travel back (gpurun merges <= 64 MiB): drop the raw rocprofv3 databases
rm -rf $O/fetch $O/write $O/st1 $O/st2 $O/st3 $O/lds0 $O/lds1 $O/prof $O/prof_cfg4 $O/prof_cfg5 $O/fetch_cfg4 $O/write_cfg4 $O/mfma_cfg4 $O/fetch_cfg5 $O/write_cfg5 $O/mfma_cfg5
for f in $O/bench_*.json $O/bench.json; do echo "$f: $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d.get('ms_per_step'), d.get('value'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'))")"; done
