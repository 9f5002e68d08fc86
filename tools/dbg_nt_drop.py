#!/usr/bin/env python3
"""dev: the bf16-mode NT GEMM epilogues with dropout against fp64 + the emulated masks (embedding shape: K = 96, position table, scale)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'nylon-amt_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
from hftt_hip import ops
import util
dev = torch.device('cuda:0'); BF = torch.bfloat16
for (M, N, K, a_bf, c_bf, table, act) in ((1024, 256, 96, False, True, True, 0), (1024, 256, 96, False, False, True, 0), (1000, 256, 96, False, True, True, 0),
                                           (1024, 512, 256, True, True, False, 1), (1024, 256, 512, True, False, False, 0), (2048, 64, 96, False, True, True, 0)):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    tab = torch.randn(32, N, generator=g) if table else None
    Ad = A.to(BF) if a_bf else A
    p, site, seed = 0.25, 5, 777
    Cd = ops.gemm_nt(Ad.to(dev), W.to(dev), b.to(dev), npass=1, act=act, out_scale=2.0, add_table=(tab.to(dev) if table else None), add_mod=32 if table else 0,
                     drop_p=p, drop_site=site, drop_seed=seed, out_dtype=BF if c_bf else torch.float32)
    ref = (Ad.double().to(BF).double() if not a_bf else Ad.double()) @ W.to(BF).double().T + b.double()
    if act == 1: ref = ref.clamp(min=0)
    ref = ref * 2.0
    if table: ref = ref + tab.double()[torch.arange(M) % 32]
    mask = util.keep_mask_t(seed, site, (M, N), p).double()
    ref = ref * mask / (1 - p)
    zeros_ok = torch.equal((Cd == 0).cpu() | (mask == 1), torch.ones(M, N, dtype=torch.bool)) and torch.equal(((Cd != 0).cpu() | (mask == 0) | (ref == 0)), torch.ones(M, N, dtype=torch.bool))
    print('M %5d N %3d K %3d a_bf %d c_bf %d table %d act %d: rel err %.4f  mask pattern ok %s' % (M, N, K, a_bf, c_bf, table, act, util.rel_err(Cd, ref), zeros_ok), flush=True)
