# every BASELINE config at full size on one GPU -> gpurun_out/${TAG}_configs_full_size_n1.jsonl   usage: bash tools/configs_round.sh TAG
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
: > gpurun_out/${TAG}_configs_full_size_n1.jsonl
for cfg in cfg1 cfg2 cfg3 cfg5; do
  timeout -k 10 400 python tools/run_config.py $cfg 2> gpurun_out/${TAG}_$cfg.err | tail -1 >> gpurun_out/${TAG}_configs_full_size_n1.jsonl || exit 1
  echo "$cfg done"
done
# the 3-D configs again with settings["preconditioner"] = "amg" (the multigrid V-cycle) -> gpurun_out/${TAG}_configs_multigrid_n1.jsonl
: > gpurun_out/${TAG}_configs_multigrid_n1.jsonl
for cfg in cfg3 cfg5; do
  timeout -k 10 300 python tools/run_config.py $cfg --preconditioner amg 2> gpurun_out/${TAG}_${cfg}_mg.err | tail -1 >> gpurun_out/${TAG}_configs_multigrid_n1.jsonl || exit 1
  echo "$cfg multigrid done"
done
# ... and the long Jacobi runs with settings["spectral_start"] = "auto" (32 vectors once a space has seen 48 solves; the harvest's seconds are inside solve_s) -> gpurun_out/${TAG}_configs_spectral_n1.jsonl
: > gpurun_out/${TAG}_configs_spectral_n1.jsonl
for cfg in cfg3 cfg5; do
  timeout -k 10 400 python tools/run_config.py $cfg --spectral-start auto 2> gpurun_out/${TAG}_${cfg}_sp.err | tail -1 >> gpurun_out/${TAG}_configs_spectral_n1.jsonl || exit 1
  echo "$cfg spectral done"
done
