"""kmpc_sim_kernel: time of one 0.1 s control period (10 model updates = 100 sub-steps) for B vehicles, by HIP events."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import _lib
if os.environ.get("KMPC_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["KMPC_LIB"])
from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
for B in (1, 4096, 262144):
    sim = VehicleSimulator(B)
    sim.state[:, 3] = 8.0; sim.state[:, 2] = torch.linspace(-3.0, 3.0, B, dtype=torch.float64, device=sim.device)
    sim._mpc_cmd_callback(0.3, 0.05)
    for _ in range(3): sim._update_vehicle_model(10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): sim._update_vehicle_model(10)
    e1.record(); torch.cuda.synchronize()
    print("B=%6d: %.1f us per control period" % (B, e0.elapsed_time(e1) / 20 * 1e3))
