import numpy as np, sys, torch
sys.path.insert(0,'/root/repo')
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
d=make_batch(4096,50,cfg_id=5)
s=BatchMPC(N=50); o=s.solve(d["z0"],d["ref"],d["v_target"],d["u_prev"]); torch.cuda.synchronize()
st=o["status"].cpu().numpy(); it=o["iters"].cpu().numpy()
print("bad idx",np.where(st!=0)[0],st[st!=0],it[st!=0], "max iters",it.max())
