"""CPU experiment (round 5): as posemb_oracle_mini.py on FLOOR-dominated clips (most bins at the log-mel floor -18.42, sparse
active bands: the shape of the plucked-string corpus).   python tests/experiments/posemb_oracle_floor.py 1 30 300 3000"""
import sys, time, math
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'tests'),os.path.join(R,'nylon-amt_amd'),R]
import torch, util
from util import O, MINI
import test_convergence_gpu as TC
torch.set_num_threads(8)
cfg=MINI
def make(cfg,n,seed):
    spec,lab=TC.make_clips(cfg,n,seed)
    # pluck-like: silent floor except where a note is on (band level stays), i.e. most bins at -18.42
    W=cfg.n_frame+2*cfg.n_margin; band=cfg.n_bin//cfg.n_note
    g=torch.Generator().manual_seed(seed)
    env=torch.randn(n,cfg.n_note,W//4+2,generator=g)
    env=torch.nn.functional.interpolate(env,size=W,mode='linear',align_corners=True)
    on=(env>0.8).repeat_interleave(band,dim=1)   # sparse activity
    s2=torch.where(on, spec[:, :cfg.n_note*band], torch.full_like(spec[:, :cfg.n_note*band], -18.420681))
    spec=spec.clone(); spec[:, :cfg.n_note*band]=s2
    lvl=env[:,:,cfg.n_margin:cfg.n_margin+cfg.n_frame].transpose(1,2)
    mpe=(lvl>0.8).float(); prev=torch.cat([mpe[:,:1],mpe[:,:-1]],1)
    onset=((mpe-prev)>0).float(); offset=((prev-mpe)>0).float()
    vel=((lvl.clamp(-2,2)+2)/4*(cfg.n_velocity-1)).round().long()*mpe.long()
    return spec.contiguous(),(onset.contiguous(),offset.contiguous(),mpe.contiguous(),vel.contiguous())
data=make(cfg,64,1); held=make(cfg,48,2)
print('active frac',float(data[1][2].mean()))
def run(scale_pos, steps=600, lr=1e-3, B=4, center=False):
    model=util.build_model(cfg,2025,dropout=0.0)
    sd={k:v.detach().clone() for k,v in model.state_dict().items()}
    for k in sd:
        if 'pos_embedding' in k: sd[k]*=scale_pos
    sd={k:v.requires_grad_(True) for k,v in sd.items()}
    opt=torch.optim.Adam(list(sd.values()),lr=lr)
    spec,labels=data; n=spec.shape[0]; acc=0; out=[]
    t0=time.time()
    for s in range(steps):
        idx=[(s*B+i)%n for i in range(B)]
        opt.zero_grad()
        loss=O.spec2midi_loss(O.model_forward(sd,spec[idx],cfg),*[t[idx] for t in labels])
        loss.backward(); opt.step(); acc+=float(loss.detach())
        if (s+1)%100==0:
            with torch.no_grad(): o=O.model_forward(sd,held[0],cfg)
            out.append((s+1, round(acc/100,4), round(TC.frame_auc(o[7],held[1][2]),3), round(TC.frame_f1(o[7],held[1][2]),3))); acc=0
    print('pos x%g lr %g: %s (%.0fs)'%(scale_pos,lr,out,time.time()-t0),flush=True)
for sc in [float(a) for a in sys.argv[1:]]: run(sc)
