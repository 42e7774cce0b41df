# PMC passes over tools/time_epilogue.py for the epilogue kernels (pica2_kernel, hfst_kernel); run on the GPU box from the repo root.
# Each counter group is its own run (counters only, with --kernel-trace); a group the device does not know is reported and skipped.
export TMPDIR=/tmp
O=gpurun_out/epi_pmc
mkdir -p $O
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
         "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if timeout -k 10 240 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O/g$i -o p -- python3 tools/time_epilogue.py > $O/g$i.out 2> $O/g$i.err; then
    echo "group $i ok: $G"
  else
    echo "group $i FAILED: $G"; tail -3 $O/g$i.err
  fi
done
python3 tools/summarise_pmc_epilogue.py $O | tee $O/summary.txt
