for a in 0 1 2 3; do
MSF_ORB_ABLATE=$a MSF_ORB_FAST_TAU=120 python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $a', d['roofline']['stage_ms'])"
done
