# SQ counter passes over the C2 bench step (two passes: the SQ block has 8 slots); summaries via tools/pmc_summary.py
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sq
KM_BENCH_NO_SPINUP=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sq/a -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-split > gpurun_out/sq/a.log 2>&1
echo pass A done
KM_BENCH_NO_SPINUP=1 timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d gpurun_out/sq/b -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-split > gpurun_out/sq/b.log 2>&1
echo pass B done
python tools/pmc_summary.py gpurun_out/sq/a > gpurun_out/sq/a.txt
python tools/pmc_summary.py gpurun_out/sq/b > gpurun_out/sq/b.txt
