cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for zc in 8 16 32 64; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcz_$zc -- python3 tools/pmc_spmv_sym.py 256 grid $zc > gpurun_out/pmcz_$zc.log 2>&1 || echo FAILED $zc
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmczt_$zc -- python3 tools/pmc_spmv_sym.py 256 grid $zc > gpurun_out/pmczt_$zc.log 2>&1 || echo FAILED $zc
done
echo finished
