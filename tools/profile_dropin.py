"""Kernel list of one tracking iteration driven through SLAMNetwork.forward + torch autograd + torch.optim.Adam (TrackingStepper,
default policies = the `dropin.pose_only` row of bench.py), graph-replayed: run under rocprofv3 --kernel-trace --stats."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nicer_slam_amd.tracking import TrackingStepper

dev = torch.device("cuda", 0)
args = argparse.Namespace(samples=128, engine="auto", precision="fp32", param_grads=True)
model, conf = bench.make_model(args, dev)
model.tracking_param_grads = False
K = torch.eye(4, device=dev)
K[0, 0] = K[1, 1] = 600.0
K[0, 2], K[1, 2] = 599.5, 339.5
gen = torch.Generator(device=dev).manual_seed(1)
batches = [bench.synth_batch(gen, 1024, dev) for _ in range(100)]
cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=dev)
st = TrackingStepper(model, K, 1024, cam, lr=0.005, use_graph=True, world=1)
for b in batches:
    st.step(*b)
torch.cuda.synchronize()
print("engine", model.last_engine)
