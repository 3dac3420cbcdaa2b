# conv8 A/B session: bash tools/conv8_ab.sh  (C8_VARIANTS="d_X_1 ..." adds library variants; C8_NO_STEP=1 skips the step-level part)
for v in $C8_VARIANTS ""; do export PTC_LIB_VARIANT=$v; echo "variant=$v"
C8_ABL=0,16 timeout 300 python tools/conv8_time.py 128 96 2>&1 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 128 128 2>&1 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 96 96 2>&1 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 96 128 2>&1 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 128 128 2 102400 | tail -1
C8_ABL=0 timeout 300 python tools/conv8_time.py 256 256 1 12115 | tail -1
done
unset PTC_LIB_VARIANT
timeout 600 python -m pytest tests -q -m gpu -x -k "block_staged_wide" 2>&1 | tail -1
[ -n "$C8_NO_STEP" ] && exit 0
for r in 1 2; do for c in 0 1; do
  echo "PTC_CONV8=$c spunet"; PTC_CONV8=$c timeout 600 python bench.py --model spunet --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-fp16-recipe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
for c in 0 1; do echo "PTC_CONV8=$c ptv3"; PTC_CONV8=$c timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-fp16-recipe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
