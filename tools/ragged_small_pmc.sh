# stall / pipe counters of the small-tile ragged kernels beside the large-tile ones (configs[4] share)
O=gpurun_out/rgs; mkdir -p $O; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
db() { ls $1/*/*.db $1/*.db 2>/dev/null | head -1; }
W="python bench.py --workload cfg5 --shard-of 8 --no-graph --no-cpu-baseline --no-roofline --no-fast-path --steps 20 --warmup 3 --min-seconds 0"
for S in 0 1; do
  V2X_RAGGED_SMALL=$S rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/a$S -o pmc -- $W > /dev/null 2> $O/a$S.err
  python tools/rocpd_summary.py $(db $O/a$S) 2>&1 | grep -e "^kernel" -e ragged > $O/pmc_a$S.txt
  V2X_RAGGED_SMALL=$S rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU -d $O/b$S -o pmc -- $W > /dev/null 2> $O/b$S.err
  python tools/rocpd_summary.py $(db $O/b$S) 2>&1 | grep -e "^kernel" -e ragged > $O/pmc_b$S.txt
  rm -rf $O/a$S $O/b$S
done
cat $O/pmc_a0.txt $O/pmc_a1.txt $O/pmc_b0.txt $O/pmc_b1.txt
