for rep in 1 2; do
for hw in 0 208 52; do
  echo "== DFE_PLANECONV_MAX_HW=$hw"; DFE_PLANECONV_MAX_HW=$hw python bench.py --steps 30 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"
done
done
