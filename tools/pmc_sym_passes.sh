# rocprofv3 --pmc passes over the three forms of the PCG product (one counter set per pass); see tools/pmc_spmv_sym.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in csr rows grid; do
  for set in "$@"; do
    tag=$(echo $set | tr ' ' '_')
    rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcs_${mode}_${tag} -- python3 tools/pmc_spmv_sym.py 256 $mode 16 > gpurun_out/pmcs_${mode}_${tag}.log 2>&1 || echo "FAILED $mode $set"
  done
done
echo finished
