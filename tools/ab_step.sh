# Alternating runs of the headline step on ONE device: `tools/ab_step.sh KNOB=VALUE [rounds] [bench args]`
# prints ms/step with the knob unset ("base") and set ("knob"), alternating, so that device-to-device spread cancels.
KV=$1; ROUNDS=${2:-3}; shift; shift
for k in $(seq $ROUNDS); do
  for arm in base knob; do
    if [ $arm = knob ]; then export "$KV"; else unset "${KV%%=*}"; fi
    timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-phases "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$arm', d['ms_per_step'], d['value'])" || exit 1
  done
done
