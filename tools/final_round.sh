# the round's last GPU call: full -m gpu suite, smoke(), bench + the same under rocprofv3 (tools/profile_round.sh), every BASELINE config (tools/configs_round.sh)   usage: bash tools/final_round.sh TAG
TAG=$1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/${TAG}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh $TAG && bash tools/configs_round.sh $TAG
F=$(find gpurun_out/prof_${TAG} -name '*kernel_stats.csv' | head -1); cp "$F" gpurun_out/${TAG}_kernel_stats.csv
python tools/kernel_stats_filtered.py gpurun_out/prof_${TAG} > gpurun_out/${TAG}_kernel_stats_filtered.csv; rm -rf gpurun_out/prof_${TAG}
cat gpurun_out/${TAG}_bench.json | cut -c1-300
