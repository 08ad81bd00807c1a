for v in kth4 kth2 kth8 kth1 kth4; do
TMX_LIBRARY=$PWD/build/abl/$v.so python3 bench.py --steps 10 --warmup 2 --develop 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s h_explicit %.3f ms/step  (step %.3f ms)' % ('$v', d['kernel_ms_per_step']['h_explicit'], d['ms_per_step']))"
done
