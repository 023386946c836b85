mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -iE "^\s*(Name|Counter)?.*(MFMA|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_WAIT_ANY|LDS_BANK|SQ_WAVES\b|SQ_INSTS_VALU\b|SQ_INST_CYCLES_VMEM|SQ_BUSY_CU)" | head -40 > $R/gpurun_out/counters.txt
cd /tmp
export PADEL_CONV_LDS_VARIANT=7
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $R/gpurun_out/pmc1 -o p -- python $R/tools/conv_bench.py --one 0 0 --reps 2 --shapes "m.P4.bneck,m.P3.bneck" > $R/gpurun_out/pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc2 -o p -- python $R/tools/conv_bench.py --one 0 0 --reps 2 --shapes "m.P4.bneck,m.P3.bneck" > $R/gpurun_out/pmc2.log 2>&1
ls $R/gpurun_out/pmc1 $R/gpurun_out/pmc2; tail -3 $R/gpurun_out/pmc1.log
