"""How fast the training trajectory of BASELINE configs[0] separates under perturbations of its own parameters, measured
with the oracle alone (CPU): `f16` = W rounded to f16 before every forward, `eps` = W perturbed by 1e-7 relative.  Prints the
worst relative loss difference at 1, 2, 3, 5, 10, 20, 30, 50, 100 free iterations after a common state (tests/test_gpu_longrun.py)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc
from videovector_amd.synth import SyntheticVideos, init_weights
orc.build()
B,C,Nn,F,D = 32,5,2,128,32
ds = SyntheticVideos(seed=1701, n_videos=50)
table = ds.table(F)
kw = dict(batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=500, negative_swap_percentage=50)
def lr_at(it, base): return base*(1+1e-3*it)**-0.75
def run(mode, base, n=400, force=100):
    smp = orc.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    W0,b0 = init_weights(1,D,F,std=0.02)
    Wa,ba = W0.copy(), b0.copy(); hWa,hba = np.zeros_like(Wa), np.zeros_like(ba)
    Wb,bb = W0.copy(), b0.copy(); hWb,hbb = np.zeros_like(Wb), np.zeros_like(bb)
    by = np.zeros(force)
    for it in range(n):
        idx = smp.next()[0]
        if it % force == 0: Wb[:] = Wa; bb[:] = ba; hWb[:] = hWa; hbb[:] = hba
        lr = lr_at(it, base)
        ra = orc.forward_backward(table, idx, Wa, ba, C_=C, Nn=Nn, want=("dW","db"))
        if mode == "f16": Wuse = Wb.astype(np.float16).astype(np.float32); tb = table
        elif mode == "eps": Wuse = Wb*(1+1e-7*np.sign(np.sin(np.arange(Wb.size).reshape(Wb.shape)*(it+1)))).astype(np.float32); tb = table
        rb = orc.forward_backward(tb, idx, Wuse, bb, C_=C, Nn=Nn, want=("dW","db"))
        for (W_,b_,hW_,hb_,r) in ((Wa,ba,hWa,hba,ra),(Wb,bb,hWb,hbb,rb)):
            orc.sgd_update(W_, r["dW"], hW_, lr, 1.0, .9, 5e-4, 1.0)
            orc.sgd_update(b_, r["db"], hb_, lr, 2.0, .9, 5e-4, 0.0)
        by[it%force] = max(by[it%force], abs(ra["loss"]-rb["loss"])/abs(ra["loss"]))
    return by
for mode in ("f16","eps"):
    for base in (1e-3,):
        by = run(mode, base)
        print(mode, base, " ".join("%.1e"%by[k] for k in (0,1,2,4,9,19,29,49,99)))
