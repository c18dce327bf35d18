mkdir -p gpurun_out/r2e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2e/gputests.txt
python bench.py > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2e/prof -o stats -- python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r2e/prof_bench.json 2> gpurun_out/r2e/prof.err
python tools/rocpd_summary.py $(ls gpurun_out/r2e/prof/*/*.db gpurun_out/r2e/prof/*.db 2>/dev/null | head -1) > gpurun_out/r2e/kernel_stats.txt 2>&1
cat gpurun_out/r2e/gputests.txt; cut -c1-600 gpurun_out/r2e/bench.json; head -12 gpurun_out/r2e/kernel_stats.txt
