cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in 0 1; do
if [ $e = 1 ]; then export V2X_NO_PACK_SCATTER=1; fi
rocprofv3 --kernel-trace --stats -d gpurun_out/tmpp$e -o stats -- python bench.py --no-cpu-baseline --no-roofline --no-edge-gather --min-seconds 0.3 > /dev/null 2>&1
python tools/rocpd_summary.py $(ls gpurun_out/tmpp$e/*/*.db gpurun_out/tmpp$e/*.db 2>/dev/null | head -1) | grep -E "k_reduce_adam|k_mlp_train"
done
rm -rf gpurun_out/tmpp*
