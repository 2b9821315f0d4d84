"""ctypes binding of libvad_amd.so (the C ABI declared in include/vad_amd.h).

The library is the only compute path of this package: if it is missing or a
call fails, a ``VadError`` is raised -- there is no CPU fallback.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VAD_AMD_LIB selects another build of the same library (kernel experiments); the default is the in-tree one
LIB_PATH = os.environ.get('VAD_AMD_LIB') or os.path.join(_HERE, 'libvad_amd.so')
CSRC = os.path.join(_HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(_HERE), 'include')
SOURCES = ['abi.hip', 'mfcc.hip', 'rnn.hip', 'densenet.hip', 'nicknet.hip', 'nicknet_stream.hip', 'overlay.hip', 'sweep.hip', 'seq.hip']
HEADERS = ['vad_internal.h', 'mfcc_device.h', 'rnn_device.h', 'nicknet_device.h']

# every symbol include/vad_amd.h declares
SYMBOLS = [
    'vad_abi_version', 'vad_last_error', 'vad_init', 'vad_mfcc_delta_f32', 'vad_mfcc_delta_slices_f32',
    'vad_net_blob_floats', 'vad_pack_net_weights',
    'vad_densenet_blob_floats', 'vad_densenet_tensor_count', 'vad_pack_densenet_weights',
    'vad_gru_forward_f32', 'vad_lstm_forward_f32', 'vad_densenet_forward_f32', 'vad_gru_forward_ws_f32', 'vad_gru_workspace_bytes',
    'vad_synth_audio_i16', 'vad_overlay_i16', 'vad_threshold_sweep_f32', 'vad_threshold_sweep_noalloc_f32',
    'vad_set_matrix_mode', 'vad_get_matrix_mode', 'vad_gru_forward_mode_f32', 'vad_lstm_forward_mode_f32', 'vad_gru_forward_rows_f32', 'vad_lstm_forward_rows_f32', 'vad_densenet_forward_rows_f32',
    'vad_nicknet_blob_floats', 'vad_nicknet_tensor_count', 'vad_pack_nicknet_weights', 'vad_nicknet_forward_f32',
    'vad_nicknet_forward_rows_f32', 'vad_nicknet_stages_f32', 'vad_nicknet_embed_f32', 'vad_net_forward_seq_f32',
]


class VadError(RuntimeError):
    pass


_lib = None


def build(verbose=False):
    """Compile the HIP sources for gfx950 into the in-tree shared library."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, 'vad_amd.h')]
    if os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # packed fp32 vector instructions are switched off: the compiler's pairing of scalar float math into
    # v_pk_* costs more issue time than the two plain instructions and adds register moves (measured: MFCC
    # kernel -4 %, DenseNet -7 %, GRU -2 %)
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops',
           '-I' + INCLUDE, '-I' + CSRC, '-o', LIB_PATH] + srcs
    if verbose:
        print(' '.join(cmd))
    # the device-only feature switch makes the host pass of every source print an 'ignoring feature' note
    proc = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    noise = "'-packed-fp32-ops' is not a recognized feature for this target"
    for line in proc.stderr.splitlines():
        if noise not in line:
            print(line, file=sys.stderr)
    if proc.returncode != 0:
        raise subprocess.CalledProcessError(proc.returncode, cmd)
    return LIB_PATH


def lib():
    """The loaded library; raises VadError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VadError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                       '(there is no CPU fallback)' % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    c_i64, c_f32p, c_void = ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p
    L.vad_abi_version.restype = ctypes.c_int
    L.vad_last_error.restype = ctypes.c_char_p
    L.vad_init.restype = ctypes.c_int
    L.vad_mfcc_delta_f32.restype = ctypes.c_int
    L.vad_mfcc_delta_f32.argtypes = [c_void, c_i64, c_i64, c_i64, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_void]
    L.vad_mfcc_delta_slices_f32.restype = ctypes.c_int
    L.vad_mfcc_delta_slices_f32.argtypes = [c_void, c_void, c_i64, c_i64, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_void]
    L.vad_net_blob_floats.restype = ctypes.c_size_t
    L.vad_net_blob_floats.argtypes = [ctypes.c_int] * 3
    L.vad_pack_net_weights.restype = ctypes.c_int
    L.vad_pack_net_weights.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 10 + [ctypes.c_void_p, ctypes.c_size_t]
    L.vad_densenet_blob_floats.restype = ctypes.c_size_t
    L.vad_densenet_blob_floats.argtypes = [ctypes.c_int]
    L.vad_densenet_tensor_count.restype = ctypes.c_int
    L.vad_densenet_tensor_count.argtypes = [ctypes.c_int]
    L.vad_pack_densenet_weights.restype = ctypes.c_int
    L.vad_pack_densenet_weights.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_size_t]
    fwd = [c_void, c_i64, c_i64, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, ctypes.c_int,
           ctypes.c_float, c_void]
    L.vad_gru_forward_f32.restype = ctypes.c_int
    L.vad_gru_forward_f32.argtypes = fwd
    L.vad_lstm_forward_f32.restype = ctypes.c_int
    L.vad_lstm_forward_f32.argtypes = fwd
    L.vad_gru_forward_ws_f32.restype = ctypes.c_int
    L.vad_gru_forward_ws_f32.argtypes = fwd[:-1] + [c_void, ctypes.c_size_t, c_void]
    L.vad_gru_forward_mode_f32.restype = ctypes.c_int
    L.vad_gru_forward_mode_f32.argtypes = fwd[:-1] + [ctypes.c_int, c_void, ctypes.c_size_t, c_void]
    L.vad_lstm_forward_mode_f32.restype = ctypes.c_int
    L.vad_lstm_forward_mode_f32.argtypes = fwd[:-1] + [ctypes.c_int, c_void]
    L.vad_net_forward_seq_f32.restype = ctypes.c_int
    L.vad_net_forward_seq_f32.argtypes = [c_void, c_i64, c_i64] + [c_void] * 8 + [ctypes.c_int, ctypes.c_int, c_void, c_void, c_void]
    L.vad_gru_workspace_bytes.restype = ctypes.c_size_t
    L.vad_gru_workspace_bytes.argtypes = [ctypes.c_int, c_i64]
    L.vad_densenet_forward_f32.restype = ctypes.c_int
    L.vad_densenet_forward_f32.argtypes = [c_void, c_i64, c_i64, c_void, ctypes.c_int, c_void, c_void,
                                           ctypes.c_int, ctypes.c_float, c_void]
    L.vad_nicknet_blob_floats.restype = ctypes.c_size_t
    L.vad_nicknet_blob_floats.argtypes = [ctypes.c_int]
    L.vad_nicknet_tensor_count.restype = ctypes.c_int
    L.vad_nicknet_tensor_count.argtypes = [ctypes.c_int]
    L.vad_pack_nicknet_weights.restype = ctypes.c_int
    L.vad_pack_nicknet_weights.argtypes = L.vad_pack_densenet_weights.argtypes
    L.vad_nicknet_forward_f32.restype = ctypes.c_int
    L.vad_nicknet_forward_f32.argtypes = L.vad_densenet_forward_f32.argtypes
    L.vad_nicknet_stages_f32.restype = ctypes.c_int
    L.vad_nicknet_stages_f32.argtypes = [c_void, c_i64, c_i64, c_void, c_void, ctypes.c_int] + [c_void] * 6
    L.vad_nicknet_embed_f32.restype = ctypes.c_int
    L.vad_nicknet_embed_f32.argtypes = [c_void, c_i64, c_void, ctypes.c_int, c_void, c_void, c_void]
    L.vad_synth_audio_i16.restype = ctypes.c_int
    L.vad_synth_audio_i16.argtypes = [c_void, c_i64, c_i64, ctypes.c_uint32, c_void]
    L.vad_overlay_i16.restype = ctypes.c_int
    L.vad_overlay_i16.argtypes = [c_void, c_i64, c_void, c_i64, ctypes.c_int, ctypes.c_double, c_void, c_void]
    L.vad_threshold_sweep_f32.restype = ctypes.c_int
    L.vad_threshold_sweep_f32.argtypes = [c_void, c_i64, c_void, c_i64, c_void, ctypes.c_int, c_void, c_void, c_void, c_void]
    L.vad_threshold_sweep_noalloc_f32.restype = ctypes.c_int
    L.vad_threshold_sweep_noalloc_f32.argtypes = [c_void, c_i64, c_void, c_i64, c_void, ctypes.c_int, c_void, c_void, c_void]
    L.vad_set_matrix_mode.restype = ctypes.c_int
    L.vad_set_matrix_mode.argtypes = [ctypes.c_int]
    L.vad_get_matrix_mode.restype = ctypes.c_int
    rows = [c_void, c_i64, c_void, c_i64, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, ctypes.c_int,
            ctypes.c_float, c_void]
    L.vad_gru_forward_rows_f32.restype = ctypes.c_int
    L.vad_gru_forward_rows_f32.argtypes = rows
    L.vad_lstm_forward_rows_f32.restype = ctypes.c_int
    L.vad_lstm_forward_rows_f32.argtypes = rows
    L.vad_densenet_forward_rows_f32.restype = ctypes.c_int
    L.vad_densenet_forward_rows_f32.argtypes = [c_void, c_i64, c_void, c_i64, c_void, ctypes.c_int, c_void, c_void,
                                                ctypes.c_int, ctypes.c_float, c_void]
    L.vad_nicknet_forward_rows_f32.restype = ctypes.c_int
    L.vad_nicknet_forward_rows_f32.argtypes = L.vad_densenet_forward_rows_f32.argtypes
    _lib = L
    return L


def init():
    """vad_init() on the current device: constant tables and LDS opt-ins, the two one-time steps that
    cannot happen inside a HIP graph capture.  Optional otherwise (the first call that needs them does it)."""
    check(lib().vad_init())


def check(rc):
    if rc != 0:
        raise VadError('libvad_amd call failed (%d): %s' % (rc, lib().vad_last_error().decode()))


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def host_ptr(a):
    """Pointer to a C-contiguous float32 numpy array (or None)."""
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(ctypes.c_void_p)
