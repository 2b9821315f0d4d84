"""Front-end-only timing: ms per 100 h of audio for the library selected by VAD_AMD_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import voice_activity_detection_amd as pkg

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
n = int(hours * 120000)
stream = pkg.synth_stream(0, n, seed=1337)
feat = torch.empty(n, 24, device='cuda')
for _ in range(2):
    pkg.mfcc_delta_stream(stream, n, 166, out=feat)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    pkg.mfcc_delta_stream(stream, n, 166, out=feat)
e1.record()
torch.cuda.synchronize()
print('%-40s %.2f ms per 100 h   checksum %.6f' % (os.path.basename(os.environ.get('VAD_AMD_LIB', 'in-tree')),
                                                 e0.elapsed_time(e1) / 5 * 100 / hours, float(feat.double().abs().mean())))
