for ne in 12 20 30; do
  python bench.py --ne $ne --steps 10 --warmup 2 --no-cpu-baseline --sustained-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']
print($ne, round(d['ms_per_step'],3), {a:round(b,4) for a,b in k.items() if b})"
done
