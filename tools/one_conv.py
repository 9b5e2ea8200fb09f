import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
dev = torch.device("cuda:0")
h, w = int(os.environ.get("CH", 288)), int(os.environ.get("CW", 480))
cin, cout = int(os.environ.get("CIN", 96)), int(os.environ.get("COUT", 96))
x = torch.rand(1, cin, h, w, device=dev) * 2 - 1
wt = torch.randn(cout, cin, 3, 3, device=dev) / 30
b = torch.randn(cout, device=dev)
out = torch.empty(1, cout, h, w, device=dev)
for _ in range(int(os.environ.get("REPS", 10))):
    hip.conv2d([x], wt, b, relu=True, out=out)
torch.cuda.synchronize()
