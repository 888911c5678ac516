set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["kernels_ms"].items()}, round(d["value"]/1e6,2))'
for B in 131072 16384 4096; do
for D in 0 1536 3072 0 1536; do
echo -n "B=$B dense=$D: "
AOC_LS_DENSE=$D python bench.py --steps 10 --warmup 1 --no-cpu-baseline --batch-per-gpu $B 2>/dev/null | python -c "$P"
done; done
