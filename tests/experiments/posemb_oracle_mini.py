"""CPU experiment (round 5): the reference's own step (oracle forward, train.py loss, torch Adam, fp32, dropout 0) on the
convergence test's clips at the mini size, with the three position tables multiplied by 1 / 30 / 300 at initialisation.  Prints
(step, mean loss, held-out ranking AUC, held-out frame-F1) per 100 steps.   python tests/experiments/posemb_oracle_mini.py"""
import sys, time, math
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'tests'),os.path.join(R,'nylon-amt_amd'),R]
import torch, util
from util import O, MINI
import test_convergence_gpu as TC
torch.set_num_threads(8)
cfg=MINI
data=TC.make_clips(cfg,64,seed=1); held=TC.make_clips(cfg,48,seed=2)
def run(scale_pos, scale_tok, steps=600, lr=1e-3, B=4):
    model=util.build_model(cfg,2025,dropout=0.0)
    sd={k:v.detach().clone() for k,v in model.state_dict().items()}
    for k in sd:
        if 'pos_embedding' in k: sd[k]*=scale_pos
        if 'tok_embedding' in k or 'conv' in k and False: sd[k]*=scale_tok
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
    print('pos x%g tok x%g lr %g: %s (%.0fs)'%(scale_pos,scale_tok,lr,out,time.time()-t0),flush=True)
print([k for k in util.build_model(cfg,2025,dropout=0.0).state_dict().keys() if 'embedding' in k])
run(1,1); run(30,1); run(300,1)
