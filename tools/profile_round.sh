# one round of driver-like measurements: bench, the same command under rocprofv3 --kernel-trace --stats, PMC passes
# usage: bash tools/profile_round.sh TAG
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-csr-section --no-general-paths > gpurun_out/${TAG}_prof_bench.json 2> gpurun_out/${TAG}_prof.err || exit 1
echo finished
