for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export SA_SIDE_WGRAD_VQVAE=1; else unset SA_SIDE_WGRAD_VQVAE; fi
    python bench.py --no-performer --no-extras --no-cpu-baseline --no-kernel-timer --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('side_wgrad=$v', d['value'], d['step_ms']['median'])"
  done
done
