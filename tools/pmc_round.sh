# rocprofv3 --pmc passes (one counter per pass, program directly after --): calibration streams of known size, then the
# PCG product.  usage: bash tools/pmc_round.sh TAG [variant]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02}; VAR=${2:-0}
for w in 8 16; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_calib_load$w -- python3 tools/pmc_calib.py $w load > gpurun_out/pmc_${TAG}_calib_load$w.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_calib_store$w -- python3 tools/pmc_calib.py $w store > gpurun_out/pmc_${TAG}_calib_store$w.log 2>&1 || exit 1
done
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${TAG}_march_$set -- python3 tools/pmc_spmv_sym.py 256 grid 16 $VAR > gpurun_out/pmc_${TAG}_march_$set.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py $TAG > gpurun_out/pmc_${TAG}_summary.json
cat gpurun_out/pmc_${TAG}_summary.json
