cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in 4 5; do
  export RT_AMD_POOL=1 RT_AMD_POOL_OCC=$o
  bash tools/pmc_one.sh $PWD/gpurun_out/pmcpool$o sqa "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" --kind 0 --level 8 --copies 1 --frames 16 --warmup 16
  bash tools/pmc_one.sh $PWD/gpurun_out/pmcpool$o sqb "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU" --kind 0 --level 8 --copies 1 --frames 16 --warmup 16
  cd ${GRAFT_REPO_ROOT:-/root/repo}
  echo "== occ $o"; python3 tools/pmc_summary.py gpurun_out/pmcpool$o k_paths
done
unset RT_AMD_POOL
bash tools/pmc_one.sh $PWD/gpurun_out/pmcpoolref sqa "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" --kind 0 --level 8 --copies 1 --frames 16 --warmup 16
bash tools/pmc_one.sh $PWD/gpurun_out/pmcpoolref sqb "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU" --kind 0 --level 8 --copies 1 --frames 16 --warmup 16
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== k_paths"; python3 tools/pmc_summary.py gpurun_out/pmcpoolref k_paths
