mkdir -p gpurun_out/r2h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2h/gputests.txt
python bench.py > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2h/prof -o stats -- python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r2h/prof_bench.json 2> gpurun_out/r2h/prof.err
python tools/rocpd_summary.py $(ls gpurun_out/r2h/prof/*/*.db gpurun_out/r2h/prof/*.db 2>/dev/null | head -1) > gpurun_out/r2h/kernel_stats.txt 2>&1
cat gpurun_out/r2h/gputests.txt; cut -c1-600 gpurun_out/r2h/bench.json; head -12 gpurun_out/r2h/kernel_stats.txt
