import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from mkz_mpc_path_follower_amd import _lib
if os.environ.get("KMPC_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["KMPC_LIB"])
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
from oracle import oracle as O
import certify as CT
N=28; B=2560
d=make_batch(B,N,cfg_id=2)
s=BatchMPC(N=N)
o=s.solve(d["z0"],d["ref"],d["v_target"],d["u_prev"],want_U=True); torch.cuda.synchronize()
r={k:v.cpu().numpy() for k,v in o.items()}
ro=O.solve_condensed_batch(O.params(N),d["z0"],d["ref"],d["v_target"],d["u_prev"],nthreads=8)
b=1369
print("GPU status",r["status"][b],"iters",r["iters"][b],"cost %.10f"%r["cost"][b],"| CPU status",ro["status"][b],"iters",ro["iters"][b],"cost %.10f"%ro["cost"][b])
rel=np.abs(r["cost"]-ro["cost"])/np.maximum(1,np.abs(ro["cost"])); print("max rel cost diff",rel.max(),"argmax",rel.argmax(),"iters differ on",(r["iters"]!=ro["iters"]).sum())
for name,U in (("GPU",r["U"]),("CPU",ro["U"])):
    c=CT.certify_batch(O,O.params(N),d,U,idx=np.array([b]))
    print(name,{k:float(v[0]) for k,v in c.items()})
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "U1369_%s.npy" % ("prev" if os.environ.get("KMPC_LIB") else "new")), r["U"][b])
