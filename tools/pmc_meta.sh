# Instruction mix / pipe utilisation of the Meta-Kernel alone (tools/meta_bench.py) by rocprofv3 PMC passes -- separate --pmc runs with
# --kernel-trace only (gpurun refuses --pmc together with the other trace domains).  Run on the GPU box through gpurun; summary in
# gpurun_out/p/meta_pmc.json (copy to profiles/).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/p
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1)); timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/p/mpmc$i -- python tools/meta_bench.py 8 5 > gpurun_out/p/mpmc$i.log 2>&1
done
python tools/pmc_summary.py 8 gpurun_out/p/mpmc1 gpurun_out/p/mpmc2 gpurun_out/p/mpmc3 > gpurun_out/p/meta_pmc.json 2> gpurun_out/p/meta_pmc.err
rm -rf gpurun_out/p/mpmc1 gpurun_out/p/mpmc2 gpurun_out/p/mpmc3
python -c "import json; d=json.load(open('gpurun_out/p/meta_pmc.json')); print(json.dumps(d.get('meta_kernel'), indent=1))"
