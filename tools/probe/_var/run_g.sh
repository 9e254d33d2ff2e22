mkdir -p gpurun_out/var
python bench.py --phase gan --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/var/gan.json
python - <<'P'
import json
d=json.load(open('gpurun_out/var/gan.json'))
print(d['ms_per_step'], d.get('step_mode'))
for k,v in d.get('kernel_families',{}).items(): print(k, v)
P
