cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/r01k_bench.json 2> gpurun_out/r01k_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1k -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r01k_prof_bench.json 2> gpurun_out/r01k_prof.err || exit 1
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_r1k_$set -- python3 tools/pmc_spmv_sym.py 256 grid 16 > gpurun_out/pmc_r1k_$set.log 2>&1 || exit 1
done
echo finished
