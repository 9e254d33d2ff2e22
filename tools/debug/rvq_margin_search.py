import sys, time; sys.path.insert(0,'oracle')
import torch, rave_oracle as O
torch.set_num_threads(8)
cfg=O.discrete_config()
sd=O.init_state_dict(cfg, seed=0, with_discriminator=False)
x=O.synthetic_batch(2,1,65536)
with torch.no_grad():
    zp=O.encoder_v2(O.pqmf_encode(x, sd["pqmf.forward_conv.weight"]), sd, cfg)
r0=float(zp.pow(2).mean().sqrt()); print("rms", r0)
best=[]
for cs in range(1500):
    sd.update(O.seeded_codebooks(cfg, cs, scale=r0))
    m=[]
    with torch.no_grad():
        zq,diff,ind,new=O.rvq_forward(zp, sd, "encoder.rvq", 16, True, margins=m)
    mm=torch.stack(m,1)
    best.append((float(mm.min()), cs, float((zp-zq).pow(2).mean().sqrt()/r0)))
best.sort(reverse=True); print(best[:6]); print(best[-3:])
