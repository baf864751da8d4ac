mkdir -p gpurun_out/r05
for b in 3072 3840 4096 4608 5376 6144 6912 7680 8192; do for f in "" "--pipeline"; do
python bench.py --batch $b --steps 100 --warmup 40 --no-cpu-baseline $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($b, '$f', round(d['value']), round(d['ms_per_step'],4), round(d['kernel_ms_per_step']['solve'],4))"
done; done
