# in-session A/B of library variants: bash tools/lib_variant_ab.sh "<variant> <variant> ..." <bench_ops sections> [pytest -k expression]
# ("default" names the default library); op timings per variant, then bench.py steps alternating, then the tests on the default library
VARS=${1:-default}; SECS=${2:-linear}; KEXP=$3
for v in $VARS; do
  [ "$v" = default ] && export PTC_LIB_VARIANT= || export PTC_LIB_VARIANT=$v
  echo "== variant=$v"; timeout 600 python tools/bench_ops.py --only $SECS 2>&1 | grep -v amdgpu.ids | cut -c1-260
done
for r in 1 2; do for v in $VARS; do
  [ "$v" = default ] && export PTC_LIB_VARIANT= || export PTC_LIB_VARIANT=$v
  echo "variant=$v ptv3 ms/step: $(timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-fp16-recipe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done; done
export PTC_LIB_VARIANT=
[ -n "$KEXP" ] && timeout 1500 python -m pytest tests -q -m gpu -x -k "$KEXP" 2>&1 | tail -3
