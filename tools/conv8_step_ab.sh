# step-level A/B of conv8 (PTC_CONV8=0 | 1), both backbones, one session, alternating
for r in 1 2; do for c in 0 1; do
  echo "PTC_CONV8=$c spunet"; PTC_CONV8=$c timeout 600 python bench.py --model spunet --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-fp16-recipe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  echo "PTC_CONV8=$c ptv3"; PTC_CONV8=$c timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-fp16-recipe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
timeout 900 python -m pytest tests -q -m gpu -x -k "spconv or spunet or block_staged" 2>&1 | tail -3
