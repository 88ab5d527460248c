# kernel timeline of the sharded driver with one rank (bench.py --dist-driver): what lies between the kernels of an iteration
# usage: bash tools/timeline_sharded.sh TAG [N] ; then python tools/timeline_gaps.py gpurun_out/TAG_trace.csv
TAG=${1:-tls}
N=${2:-128}
BENCH_ARGS="--dist-driver --n $N --no-general-paths" bash tools/timeline_pass.sh $TAG
