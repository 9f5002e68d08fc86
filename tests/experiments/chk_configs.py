"""dev: x3 / exact-fp32 modes against the CPU checker on configurations the test-suite does not use (generic x3 path at d = 256 with ff = 1024,
d = 128, odd batch), dropout 0: every output tensor's max |difference| and magnitude."""
import sys, os, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'nylon-amt_amd'))
import util
from util import O, OUT_NAMES
dev = torch.device('cuda:0')
mk = lambda **kw: O.HfttConfig(n_margin=4, n_frame=16, n_bin=32, cnn_channel=4, cnn_kernel=5, n_note=8, n_velocity=16, **kw)
for name, cfg in (('d256_pf1024', mk(hid_dim=256, pf_dim=1024, enc_layer=1, dec_layer=2, enc_head=4, dec_head=4)),
                  ('d128_pf256', mk(hid_dim=128, pf_dim=256, enc_layer=2, dec_layer=1, enc_head=4, dec_head=2)),
                  ('d256_pf512_B3', mk(hid_dim=256, pf_dim=512, enc_layer=1, dec_layer=1, enc_head=4, dec_head=4))):
    B = 3
    x = O.synth_spec(B, cfg, salt=31)
    model = util.build_model(cfg, 2024, dropout=0.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = O.model_forward(sd, x, cfg)
    model = model.to(dev).eval()
    for mode in ('parity', 'x3'):
        model.hftt_precision = mode
        with torch.no_grad():
            out = model(x.to(dev))
        print(name, mode, ' '.join('%s %.1e/%.0e' % (n[:6], float((o.cpu() - r).abs().max()), float(r.abs().max())) for n, o, r in zip(OUT_NAMES, out, ref)), flush=True)
