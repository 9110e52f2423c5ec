cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  REPS=1 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_$tag -o out --output-format csv -- python3 $R/tools/prof_scan.py > $R/gpurun_out/pmc_$tag.log 2>&1 || exit 1
done
echo done
