# Select-kernel timing diagnostics (AZSP_DEBUG_SELECT bits: 8 no speculative prefetch, 16 / 32 observation planes twice,
# 64 arg-max twice, 128 rules step twice -- results stay bit-exact, the launch grows by the cost of the repeated part)
for d in 0 8 16 64 128; do
  AZSP_DEBUG_SELECT=$d python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fp32 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);e=d['engine_roofline'];print('dbg=$d','select',e['avg_launch_ms'],'backup',e['backup_kernels']['avg_ms'],'nodes/sim',d['select_nodes_per_sim'],'hit',d['select_hint_hit_rate'],'moves/s',d['value'])"
done
