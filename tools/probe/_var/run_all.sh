mkdir -p gpurun_out/var
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --phase gan --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/var/gan.json
python - <<'P'
import json
d=json.load(open('gpurun_out/var/gan.json'))
print('gan ms', d['ms_per_step'], d.get('step_mode'))
for k,v in d.get('kernel_families',{}).items(): print(k, v)
P
