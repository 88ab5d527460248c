# where the HOST spends a fixed-point pass of the bench (cProfile over the whole bench.py process; top functions by own time)
cd $GRAFT_REPO_ROOT
python -m cProfile -o /tmp/bench.prof bench.py --steps 20 --warmup 5 --no-general-paths --no-pmc --no-cpu-baseline --no-csr-section > /tmp/bench_prof.json 2>/dev/null
python - <<'PY'
import pstats
p = pstats.Stats("/tmp/bench.prof")
p.sort_stats("tottime").print_stats(45)
PY
