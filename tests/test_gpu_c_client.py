"""The C ABI used from plain C (tests/c_client/vad_client.c, built with gcc against include/vad_amd.h):
the same stream and weights through the Python host side must give the same posteriors bit for bit."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest
import torch

from oracle import models_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENT = os.path.join(ROOT, 'tests', 'c_client', 'vad_client')


def build_client():
    """gcc, not hipcc: the header and the client are C99; only the HIP runtime API header is needed."""
    src = CLIENT + '.c'
    lib_dir = os.path.join(ROOT, 'voice_activity_detection_amd')
    if os.path.exists(CLIENT) and os.path.getmtime(CLIENT) >= max(os.path.getmtime(src),
                                                                   os.path.getmtime(os.path.join(ROOT, 'include', 'vad_amd.h'))):
        return CLIENT
    subprocess.run(['gcc', '-std=c99', '-Wall', '-D__HIP_PLATFORM_AMD__', '-I' + os.path.join(ROOT, 'include'),
                    '-I/opt/rocm/include', src, '-o', CLIENT, '-L' + lib_dir, '-lvad_amd', '-L/opt/rocm/lib',
                    '-lamdhip64', '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,/opt/rocm/lib'], check=True)
    return CLIENT


def test_header_is_plain_c():
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only', '-x', 'c',
                    os.path.join(ROOT, 'include', 'vad_amd.h')], check=True)


def test_client_builds_without_a_gpu():
    import voice_activity_detection_amd as pkg
    pkg.build()
    assert os.path.exists(build_client())


@pytest.mark.gpu
def test_c_client_matches_python_host_side(vad, tmp_path):
    ref = models_ref.make_model('gru_large')
    net = vad.Net(large=True, lstm=False)
    net.load_state_dict(ref.state_dict())
    blob = tmp_path / 'blob.f32'
    net._pack().tofile(str(blob))
    n, seed = 5000, 4242
    out = subprocess.run([build_client(), str(blob), str(n), str(seed)], check=True, capture_output=True, text=True,
                         timeout=120).stdout
    m = re.match(r'windows (\d+) sum_p1 (\S+) speech (\d+)((?: [0-9a-f]{8})+)', out.strip())
    assert m, out
    stream = vad.synth_stream(0, n, seed=seed)
    _, p, dec = vad.vad_stream(net, stream, n)
    p = p.cpu().numpy()
    assert int(m.group(1)) == n - 29 == p.shape[0]
    assert int(m.group(3)) == int(dec.sum())
    assert float(m.group(2)) == float(np.sum(p[:, 1].astype(np.float64)))          # same order of summation
    idx = list(range(0, n - 29, (n - 30) // 4))
    bits = [struct.unpack('<I', struct.pack('<f', float(p[w, 1])))[0] for w in idx]
    assert [int(x, 16) for x in m.group(4).split()] == bits
