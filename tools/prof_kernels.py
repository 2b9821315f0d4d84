"""Small driver for rocprofv3 counter passes: a few launches of the front end and one model kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import voice_activity_detection_amd as pkg
from bench import make_net

kind = sys.argv[1] if len(sys.argv) > 1 else 'gru_large'
hours = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.cuda.set_device(0)
net = make_net(kind)
n = int(hours * 120000)
stream = pkg.synth_stream(0, n, seed=1337)
ws = pkg.StreamWorkspace(n, stream.device)
for _ in range(reps):
    pkg.vad_stream(net, stream, n, workspace=ws)
torch.cuda.synchronize()
print('done', n, float(ws.p[:, 1].mean()))
