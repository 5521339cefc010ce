cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for layer in "encoder.model.15.conv 2000" "encoder.model.12.conv 10000" "decoder.model.6.convtr 2000 elu" "decoder.model.9.convtr 10000 elu"; do
  tag=$(echo $layer | tr ' .' '__')
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/pmc_$tag -o out --output-format csv -- python $R/tools/ablate_layer.py $layer > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc2_$tag -o out --output-format csv -- python $R/tools/ablate_layer.py $layer > /dev/null 2>&1
done
ls $R/gpurun_out/pmc_*/ | head
