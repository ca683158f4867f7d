"""How many rows of GaussianCity's hash-grid table (D5 L16 C8, 2^19 rows per level: 8 388 608 x 8 floats = 268 MB) does ONE training
step touch?  16 384 points x 32 corners x 16 levels = as many contributions as the table has rows.  Decides whether a compacted
(index, value) gradient exchange can beat DDP's dense all-reduce of the table gradient (missing item 3; DESIGN.md
section 11).  CPU only (the oracle's backward): python tools/hash_grid_touched_rows.py"""
import sys, math, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiancity_amd.grid_encoder import GridEncoder
from oracle import grid_oracle as GO
# GaussianCity BG generator: D5 L16 C8, 2^19 rows per level
enc = GridEncoder(in_channels=5, n_levels=16, lvl_channels=8, desired_resolution=2048, log2_hashmap_size=19)
print('table rows', enc.embeddings.shape, 'offsets', enc.offsets[:4].tolist(), '...', int(enc.offsets[-1]))
B=16384
rng=np.random.default_rng(0)
for name,x in (('uniform', rng.uniform(0,1,(B,5)).astype(np.float32)),
               ('one building: a 5 % box of the xyz range, two free latent coordinates', np.concatenate([0.4+0.05*rng.uniform(0,1,(B,3)), rng.uniform(0,1,(B,2))],1).astype(np.float32))):
    offs=enc.offsets.numpy().astype(np.int32)
    S=float(enc.per_level_scale)
    H=int(enc.base_resolution)
    grad=np.ones((16,B,8),np.float32)
    g=GO.backward(grad, x, tuple(enc.embeddings.shape), offs, S, H)
    g=g[0] if isinstance(g,tuple) else g
    touched=(np.abs(g).sum(1)>0)
    per=[int(touched[offs[l]:offs[l+1]].sum()) for l in range(16)]
    print(name, 'touched rows', int(touched.sum()), 'of', touched.size, '= %.1f %%'%(100*touched.mean()), 'per level', per)
