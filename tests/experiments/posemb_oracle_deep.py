"""CPU experiment (round 5): as posemb_oracle_mini.py at the paper's width and depth (d 256, ff 512, 3+3 layers, 4 heads) on short
axes.   python tests/experiments/posemb_oracle_deep.py <position-table scale> <steps> <lr>"""
import sys, time, math
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0]=[os.path.join(R,'tests'),os.path.join(R,'nylon-amt_amd'),R]
import torch, util
from util import O, MINI
import test_convergence_gpu as TC
torch.set_num_threads(8)
print(MINI)
cfg=O.HfttConfig(n_margin=4, n_frame=16, n_bin=32, cnn_channel=4, cnn_kernel=5, hid_dim=256, pf_dim=512, enc_layer=3, dec_layer=3, enc_head=4, dec_head=4, n_note=8, n_velocity=16)
data=TC.make_clips(cfg,64,seed=1); held=TC.make_clips(cfg,48,seed=2)
def run(scale_pos, steps, lr=3e-4, B=4, every=50):
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
        if (s+1)%every==0:
            with torch.no_grad(): o=O.model_forward(sd,held[0],cfg)
            out.append((s+1, round(acc/every,4), round(TC.frame_auc(o[7],held[1][2]),3), round(TC.frame_f1(o[7],held[1][2]),3))); acc=0
            print(out[-1], '%.0fs'%(time.time()-t0), flush=True)
run(float(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]))
