"""Model-kernel-only timing (stride-1 windows over random features) for the library selected by VAD_AMD_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import voice_activity_detection_amd as pkg
from bench import make_net

kind = sys.argv[1] if len(sys.argv) > 1 else 'gru_large'
mode = sys.argv[2] if len(sys.argv) > 2 else 'f32'
hours = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
n = int(hours * 120000)
net = make_net(kind)
pkg.set_matrix_mode(mode)
feat = torch.randn(n, 24, device='cuda') * 8
p = torch.empty(n - 29, 2, device='cuda')
d = torch.empty(n - 29, dtype=torch.uint8, device='cuda')
for _ in range(2):
    net.score(feat, window_stride=24, out=p, decisions=d)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    net.score(feat, window_stride=24, out=p, decisions=d)
e1.record()
torch.cuda.synchronize()
print('%-28s %-10s %-7s %.2f ms per 100 h   mean p1 %.6f' % (os.path.basename(os.environ.get('VAD_AMD_LIB', 'in-tree')), kind, mode,
                                                    e0.elapsed_time(e1) / 5 * 100 / hours, float(p[:, 1].double().mean())))
