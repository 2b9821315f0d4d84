#!/usr/bin/env python3
"""Throughput of the VAD hot path on MI355X: audio-hours processed per second.

One "step" is one pass of the whole path over one batch of synthetic audio that
is already resident in HBM: int16 frames -> MFCC+delta (166-frame slices, as
process_test_data cuts them) -> stride-1 30-frame windows -> GRU-30k forward ->
posterior + decision per window, all left in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--hours H] [--model gru_large]

For N > 1 there is one rank per GPU.  Started under torch.distributed.run the
process is a rank; started plainly (`python bench.py --gpus N`) it starts the N
ranks itself as child processes, before anything touches a GPU, relays rank 0's
JSON line and exits with the children's status.  Every rank scores its own
contiguous shard of the stream (weak scaling, no data-path collective); the
timed region is bracketed by a barrier and the slowest rank's time is used.
Rank 0 prints ONE JSON line: the headline workload (GRU-30k, 100 h per GPU)
plus, under "configs", the other model configurations timed the same way.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_HOUR = 120000          # 3600 s / 0.03 s
T, F = 30, 24                     # steps per window, features per row
PEAK_FP32_TFLOPS = 157.3          # dense fp32 matrix = fp32 vector (FMA) peak, MI355X
PEAK_BF16_TFLOPS = 2500.0         # dense bf16 matrix peak
PEAK_HBM_GBS = 8000.0
SIMDS = 256 * 4
VALU_ISSUE_CYCLES = 2.0           # a wave64 vector instruction occupies its SIMD-32 for 2 cycles
VALU_ISSUE_NS_MEASURED = 1.18                  # ns per wave64 v_add_f32 per SIMD with three waves on it (probe)
FRONTEND_BYTES_PER_FRAME = 480 * 2 + 24 * 4          # int16 in, 24 floats out
FRONTEND_FLOP_PER_FRAME = 64_000                       # 2048-pt rfft + power + mel + DCT + delta
RESULT_BYTES_PER_WINDOW = 9                            # 2 float posteriors + 1 decision byte
MODELS = ['gru_large', 'gru_small', 'lstm_large', 'lstm_small', 'gru_h90', 'densenet', 'densenet_large', 'nicknet',
          'nicknet_small']


def _rnn_flops(gates, hidden, head_rows, gate_ops):
    """(flop per window with the input projection done once per feature row, PyTorch-equivalent flop per window).
    Stride-1 windows share their rows, so W_ih x_row is algorithmically needed once per row = once per window
    of the stream; a framework that scores (B,30,24) batches computes it 30 times per window."""
    proj = F * gates * hidden * 2
    rest = T * hidden * gates * hidden * 2 + T * hidden * gate_ops + T * hidden * head_rows * 2 + \
        (head_rows * 2 * 2 if head_rows > 2 else 0)
    return proj + rest, T * proj + rest


def _densenet_flops(c0, layers, growth):
    """The same pair for DenseNet: the stem convolution (K = 24 x 6) is a property of the feature row and is
    algorithmically needed once per row; a batch framework evaluates it at 10 positions per window."""
    c1 = c0 + layers * growth
    stem = c0 * F * 6 * 2
    dense = sum((c0 + i * growth) * growth * 3 * 2 for i in range(layers))
    rest = dense * 5 + c1 * c0 * 2 * 5 + dense * 2 + c1 * c1 * 2 * 2 + c1 * 2 * 2
    return stem + rest, 10 * stem + rest


def _nicknet_flops(chans, hidden, layers, widths):
    """The same pair for the conv + GRU hybrid.  A gated convolution layer l (1-based) of a stride-1 stream is a
    property of the feature row wherever its receptive field stays clear of the window's zero padding: positions
    l .. 29 - l; only the 2 l positions next to the padding are the window's own.  Algorithmically a window
    therefore needs 2 l + 1 positions of layer l (its own, and one new row of the shared interior); a batch
    framework evaluates all 30."""
    per_pos = [2 * chans[l + 1] * chans[l] * 3 * 2 for l in range(len(chans) - 1)]
    gru = sum(3 * hidden * ((chans[-1] if g == 0 else hidden) + hidden) * 2 * T for g in range(layers)) + \
        layers * T * hidden * 12
    cols = [layers * hidden] + list(widths) + [2]
    head = sum(cols[i] * cols[i + 1] * 2 for i in range(len(cols) - 1))
    return sum((2 * (l + 1) + 1) * c for l, c in enumerate(per_pos)) + gru + head, T * sum(per_pos) + gru + head


FLOP_PER_WINDOW = {               # model -> (algorithmic for stride-1 windows, PyTorch-equivalent per (30,24) window)
    'gru_large': _rnn_flops(3, 30, 26, 12),
    'gru_small': _rnn_flops(3, 30, 2, 12),
    'lstm_large': _rnn_flops(4, 30, 26, 15),
    'lstm_small': _rnn_flops(4, 30, 2, 15),
    'gru_h90': _rnn_flops(3, 90, 26, 12),          # enlarged GRU, ~100k parameters
    'densenet': _densenet_flops(24, 6, 3),
    'densenet_large': _densenet_flops(48, 8, 4),
    'nicknet': _nicknet_flops((F, 32, 28, 25, 18), 29, 2, (32, 18)),      # the reference's conv + GRU hybrid, ~30k
    'nicknet_small': _nicknet_flops((F, 20, 18, 16), 21, 1, (18,)),      # ~10k parameters
}
# the kernel that scores stride-1 windows of each model, as rocprofv3 names it (template arguments included)
MODEL_KERNEL = {'gru_large': 'rnn_stream_kernel<false, true, 15>', 'gru_small': 'rnn_stream_kernel<false, false, 15>',
                'lstm_large': 'rnn_stream_kernel<true, true, 15>', 'lstm_small': 'rnn_stream_kernel<true, false, 15>',
                'gru_h90': 'gru_wide_stream_kernel<true, 3, 13>',
                'densenet': 'densenet_stream_kernel<false>', 'densenet_large': 'densenet_stream_kernel<true>',
                'nicknet': 'nicknet_stream_kernel<true>', 'nicknet_small': 'nicknet_stream_kernel<false>'}


DEFAULT_MODE = 'f16x6'
MATRIX_MODE_NAMES = ['f32', 'bf16x6', 'bf16x8', 'bf16x9', 'f16x6', 'f16x4']
HIDDEN = {'gru_h90': 90}                     # every other recurrent model has the reference's 30 units


def split_mode_applies(model):
    """The bf16 split matrix modes apply to the stride-1 GRU / LSTM kernels (30 units, and the wide GRU's
    workspace path); DenseNet always computes on the float32 MFMA."""
    return not MODEL_KERNEL[model].startswith(('densenet', 'nicknet'))


def model_peak(model, mode):
    """The float32 flop rate the hardware could sustain on a model's algorithmic work.  float32 MFMA and float32
    vector work share one pipe (157.3 TF).  In the bf16 split modes the recurrent and 26-row head products run as
    9, 8, 6 or 4 bf16 / f16 partial products each on the 16-bit matrix pipe, beside the rest of the work (projection, gate
    math, 2-row head) on the float32 pipe: the slower of the two pipes bounds it."""
    if model.startswith('nicknet'):
        # whatever the mode: the products of a hidden state (recurrent, and the second layer's input) are six float16
        # partial products on the 16-bit pipe, the rest is float32 MFMA / vector work
        hidden, layers = (29, 2) if model == 'nicknet' else (21, 1)
        total = FLOP_PER_WINDOW[model][0]
        on_f16 = T * 3 * hidden * hidden * 2 * (2 * layers - 1)
        seconds = max(on_f16 * 6 / (PEAK_BF16_TFLOPS * 1e12), (total - on_f16) / (PEAK_FP32_TFLOPS * 1e12))
        return total / seconds / 1e12
    if mode == 'f32' or not split_mode_applies(model):
        return PEAK_FP32_TFLOPS
    gates = 4 if model.startswith('lstm') else 3
    hidden = HIDDEN.get(model, 30)
    head_rows = 2 if model.endswith('small') else 26            # the 2-row head stays on the vector ALU
    total = FLOP_PER_WINDOW[model][0]
    on_bf16 = T * (hidden * gates * hidden * 2 + (hidden * head_rows * 2 if head_rows == 26 else 0))
    seconds = max(on_bf16 * int(mode[-1]) / (PEAK_BF16_TFLOPS * 1e12), (total - on_bf16) / (PEAK_FP32_TFLOPS * 1e12))
    return total / seconds / 1e12


def model_kernel(model, mode):
    """Kernel name for a model under a matrix mode, as rocprofv3 prints it."""
    name = MODEL_KERNEL[model]
    if mode != 'f32' and name.startswith('rnn_stream_kernel'):
        lstm, large = name[name.index('<') + 1:].split(', ')[:2]
        return 'rnn_stream_split_kernel<%s, %s, %s, %s>' % (lstm, large, mode[-1], 'true' if mode.startswith('f16') else 'false')
    if mode != 'f32' and name.startswith('gru_wide_stream_kernel'):
        large, ut = name[name.index('<') + 1:].split(', ')[:2]
        return 'gru_wide_split_kernel<%s, %s, %s, 1, %s>' % (large, ut, mode[-1], 'true' if mode.startswith('f16') else 'false')
    return name


# ---------------------------------------------------------------------------------------------
# parent mode: start one rank per GPU

def spawn_ranks(n, argv):
    """Run this script under torch.distributed.run with n local ranks and relay rank 0's line.
    Nothing in this process has touched a GPU (torch is not even imported yet)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.lstrip().startswith('{'):
            sys.stdout.write(line)
        else:
            sys.stderr.write(line)
        sys.stdout.flush()
    return proc.wait()


# ---------------------------------------------------------------------------------------------
# models and the CPU baseline

def make_net(kind):
    """The product model with seeded random weights (no checkpoints ship): torch's default initialisation of
    the modules under the reference's seed (core/train.py:56), BatchNorm statistics perturbed away from the
    identity.  Nothing from oracle/ is involved in the measured path."""
    import torch
    import voice_activity_detection_amd as pkg
    torch.manual_seed(1337)
    if kind == 'gru_h90':                              # no class in the reference: GRU(24, 90) + the same heads
        return pkg.Net(large=True, lstm=False, hidden=90)
    if kind.startswith(('densenet', 'nicknet')):
        net = pkg.DenseNet(large=kind.endswith('large')) if kind.startswith('densenet') else \
            pkg.NickNet(large=not kind.endswith('small'))
        g = torch.Generator().manual_seed(1338)
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    n = m.num_features
                    m.running_mean.copy_(torch.randn(n, generator=g) * 0.5)
                    m.running_var.copy_(torch.rand(n, generator=g) * 1.5 + 0.5)
                    m.weight.copy_(torch.rand(n, generator=g) + 0.5)
                    m.bias.copy_(torch.randn(n, generator=g) * 0.3)
        return net
    rnn, size = kind.split('_')
    return pkg.Net(large=(size == 'large'), lstm=(rnn == 'lstm'))


def make_models(kind):
    """(product model, CPU oracle model with the same weights): for the checker-side tools and the CPU baseline."""
    from oracle import models_ref, nicknet_ref
    net = make_net(kind)
    if kind.startswith('nicknet'):
        ref = nicknet_ref.NickNetRef(large=not kind.endswith('small'))
    else:
        ref = models_ref.make_model('gru_large', hidden=90) if kind == 'gru_h90' else models_ref.make_model(kind)
    ref.load_state_dict(net.state_dict())
    return net, ref.eval()


def cpu_baseline(kind, budget_s=14.0):
    """The CPU oracle (float64 numpy/scipy MFCC + torch CPU model, batch 2048) on a bounded sample of the same
    workload, once per thread count in {1, 4, 8, 16} (capped by the box's CPU share).  torch's intra-op threads
    and the BLAS / OpenMP pools are limited TOGETHER for each leg, so a leg never runs more threads than it is
    labelled with.  `value` is the fastest leg, `cores` its thread count; every leg is reported."""
    import torch
    from oracle import pipeline_ref, synth_ref
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                                # BLAS keeps its own default then
        from contextlib import nullcontext

        def threadpool_limits(limits=None):
            return nullcontext()
    _, ref = make_models(kind)
    minute = synth_ref.synth_frames(0, 2000, seed=1337)            # 1 min = 2000 frames
    pipeline_ref.vad_stream(ref, minute[:400])                     # warm-up

    def leg(budget):
        t0 = time.perf_counter()
        reps = 0
        while True:
            pipeline_ref.vad_stream(ref, minute)
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= budget or reps >= 200:
                return reps, dt

    before = torch.get_num_threads()
    share = max(1, min(int(os.environ.get('OMP_NUM_THREADS', '0') or 0) or before, os.cpu_count() or 1))
    counts = sorted({t for t in (1, 4, 8, 16) if t <= share} | {1})
    legs = {}
    for threads in counts:
        torch.set_num_threads(threads)
        with threadpool_limits(limits=threads):
            reps, dt = leg(budget_s / len(counts))
        legs[threads] = {'value': reps * 2000 / FRAMES_PER_HOUR / dt, 'reps': reps, 'seconds': dt}
    torch.set_num_threads(before)
    best = max(legs, key=lambda k: legs[k]['value'])
    return {'value': legs[best]['value'], 'unit': 'audio-hours/s', 'cores': best, 'kind': 'port',
            'by_threads': {str(k): v['value'] for k, v in legs.items()},
            'legs': {str(k): v for k, v in legs.items()}, 'host_cpus': os.cpu_count(),
            'sample': '%d x 1 min synthetic audio, f64 numpy MFCC + torch CPU %s batch 2048; best of %s threads: %d'
                      % (legs[best]['reps'], kind, '/'.join(str(k) for k in counts), best)}


# ---------------------------------------------------------------------------------------------
# HBM traffic and instruction counts measured by rocprofv3 --pmc passes (tools/pmc_summary.py --json)

def csrc_hash():
    """Identifies the kernel sources a counter profile belongs to."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, 'voice_activity_detection_amd', 'csrc')
    for name in sorted(os.listdir(base)):
        if name.endswith(('.hip', '.h')):
            with open(os.path.join(base, name), 'rb') as f:
                h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def load_pmc_profile():
    """profiles/pmc_traffic.json if it was collected on this build of the kernels, else None: counters cannot
    be read from inside this process, so a profile of other sources must not be reported as this run's."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            prof = json.load(f)
    except (OSError, ValueError):
        return None
    return prof if prof.get('csrc_hash') == csrc_hash() else None


def pmc_per_unit(prof, kernel_name):
    """Per-frame (per-window) figures of one kernel from the profile: HBM bytes and VALU wave-instructions."""
    if not prof:
        return None
    for name, k in prof.get('kernels', {}).items():
        if name == kernel_name and k.get('units'):
            out = {'file': 'profiles/pmc_traffic.json', 'kernel': name}
            if 'hbm_bytes' in k:
                out['bytes_per_unit'] = k['hbm_bytes'] / k['units']
            if 'SQ_INSTS_VALU' in k:
                out['valu_insts_per_unit'] = k['SQ_INSTS_VALU'] / k['units']
            if 'SQ_INSTS_MFMA' in k:
                out['mfma_insts_per_unit'] = k['SQ_INSTS_MFMA'] / k['units']
            return out
    return None



# ---------------------------------------------------------------------------------------------
# the line the driver parses: short (the driver keeps a few KB of output), the full record goes to a file

def _r(x, digits=5):
    """Floats to `digits` significant digits (the compact line only; the detail file keeps everything)."""
    if isinstance(x, float):
        return float('%.*g' % (digits, x))
    return x


def compact_record(out):
    """The one JSON object printed on stdout, from the full record: contract keys, ONE roofline object, the
    front end's three yardsticks, one figure per extra configuration and matrix mode, the CPU baseline.
    No notes, no per-kernel duplicates; every string stays short."""
    def roof(k):
        keep = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes', 'ms_per_launch')
        r = {key: _r(k.get(key)) for key in keep}
        if 'fp32_frac' in k:        # the front end is vector-ALU bound: its fraction of the fp32 peak is the binding one
            r['frac_fp32'] = _r(k['fp32_frac'])
        return r

    c = {key: _r(out[key]) for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                                       'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'dtype_note', 'matrix_mode',
                                       'data', 'dist_backend')}
    modes = out.get('matrix_modes') or {}
    if 'f32' in modes:                                  # the strict float32 (fp32 MFMA) figure beside the headline
        c['value_f32_mfma'] = _r(modes['f32']['value'])
        c['ms_per_step_f32_mfma'] = _r(modes['f32']['ms_per_step'])
    if 'bf16x8' in modes and out.get('matrix_mode') != 'bf16x8':      # the previous rounds' default, for comparison
        c['value_bf16x8'] = _r(modes['bf16x8']['value'])
    c['config'] = out['config']
    c['x_realtime_per_gpu'] = _r(out['x_realtime_per_gpu'])
    c['roofline'] = roof(out['roofline'])
    m, fe = out['kernels']['model'], out['kernels']['frontend']
    c['model'] = {'kernel': m['kernel'], 'ms': _r(m['ms_per_launch']), 'achieved': _r(m['achieved']),
                  'peak': _r(m['peak']), 'unit': m['unit'], 'frac': _r(m['frac']),
                  'frac_of_fp32_peak': _r(m['frac_of_fp32_peak']), 'traffic': _r(m['traffic'])}
    issue = fe.get('valu_issue') or {}
    hbm_priced = fe.get('bound') == 'hbm'                   # records written before the front end was priced as compute
    c['frontend'] = {'kernel': fe['kernel'], 'ms': _r(fe['ms_per_launch']),
                     'GBs': _r(fe['achieved'] if hbm_priced else fe['hbm_GBs']),
                     'frac_hbm': _r(fe['frac'] if hbm_priced else fe['hbm_frac']), 'frac_fp32': _r(fe['fp32_frac']),
                     'frac_issue': _r(issue.get('frac')), 'traffic': _r(fe['traffic'])}
    if out.get('configs'):
        c['configs'] = {name: {'value': _r(rec['value']), 'ms_per_step': _r(rec['ms_per_step']),
                               'model_ms': _r(rec['kernels']['model']['ms_per_launch']),
                               'frac': _r(rec['kernels']['model']['frac'])}
                        for name, rec in out['configs'].items()}
    if modes:
        c['matrix_modes'] = {mm: {'value': _r(v['value'], 4), 'model_ms': _r(v['model_kernel_ms'], 4),
                                  'max_diff': _r(v['max_abs_posterior_diff_vs_f32'], 2)}
                             for mm, v in modes.items()}
    if out.get('cpu_baseline'):
        b = out['cpu_baseline']
        c['cpu_baseline'] = {'value': _r(b['value']), 'unit': b['unit'], 'cores': b['cores'], 'kind': b['kind'],
                             'by_threads': {k: _r(v) for k, v in b['by_threads'].items()}, 'sample': b['sample']}
    if out.get('detail'):
        c['detail'] = out['detail']
    return c


COMPACT_LIMIT = 3000      # characters; the driver keeps only a few KB of a run's output


def compact_line(out):
    """compact_record as one JSON line under COMPACT_LIMIT characters: optional parts are dropped, least
    important first, rather than letting the line outgrow what the driver reads."""
    c = compact_record(out)
    for optional in (None, 'matrix_modes', 'configs', 'model', 'frontend'):
        if optional:
            c.pop(optional, None)
        line = json.dumps(c)
        if len(line) < COMPACT_LIMIT:
            return line
    raise ValueError('compact bench record is %d characters' % len(line))


def write_detail(out):
    """The full record (per-kernel rooflines with their notes, every configuration) as a file next to the
    GPU run's other outputs; returns the path relative to the repository, or None if it cannot be written."""
    rel = os.environ.get('BENCH_DETAIL', os.path.join('gpurun_out', 'bench_detail.json'))
    path = os.path.join(ROOT, rel)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(out, f, indent=1)
    except OSError:
        return None
    return rel

# ---------------------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--hours', type=float, default=100.0, help='audio hours per GPU per step')
    ap.add_argument('--model', default='gru_large', choices=MODELS)
    ap.add_argument('--slice', type=int, default=166, help='frames per MFCC slice (reference test path: 166)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--no-configs', action='store_true', help='time the headline workload only')
    ap.add_argument('--matrix-mode', default=DEFAULT_MODE, choices=MATRIX_MODE_NAMES,
                    help="how the stride-1 GRU/LSTM kernels form float32 products (default: the library's default, "
                         'f16x6 = float32 operands split into three float16 on exact power-of-two scales, the six '
                         'largest partial products accumulated in float32 on the f16 matrix pipe: under 2^-35 of a '
                         'product is left out)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # BENCH_FORCE_DIST=1: a single rank still creates its process group and goes through every collective of the
    # multi-rank path (barrier, MAX / SUM / MIN all_reduce, destroy), so that path can be run on a one-GPU box
    force_dist = os.environ.get('BENCH_FORCE_DIST', '') not in ('', '0')
    if force_dist and 'MASTER_ADDR' not in os.environ:
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(s.getsockname()[1]), RANK=str(rank),
                          WORLD_SIZE=str(world), LOCAL_RANK=str(local))
        s.close()
    args.gpus = world
    # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share a GPU when rehearsing the
    # multi-rank path on a one-GPU box (timing is then meaningless, the control flow is not)
    backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
    local = local % max(torch.cuda.device_count(), 1) if backend == 'gloo' else local
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    red_dev = 'cuda' if backend == 'nccl' else 'cpu'

    import voice_activity_detection_amd as pkg
    from voice_activity_detection_amd.sharding import shard_plan
    pkg.init()
    pkg.set_matrix_mode(args.matrix_mode)
    current_mode = [args.matrix_mode]
    prof = load_pmc_profile()

    def barrier():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_workload(model, hours, steps, warmup, buffers=None):
        """Time `steps` passes of the path for `model` over this rank's shard of world x hours of audio.
        Returns (record for rank 0, buffers for reuse by a workload of the same size)."""
        frames_per_gpu = int(round(hours * FRAMES_PER_HOUR))
        # contiguous slice-aligned shard of one stream; the rank re-creates its own lead-in and halo,
        # and scores exactly the windows that start inside its shard (no inter-rank traffic)
        plan = shard_plan(frames_per_gpu * world, world, rank, args.slice)
        n_frames, n_windows = plan['n_load'], plan['n_windows']
        if buffers is None or buffers[0] != (plan['first'], n_frames):
            buffers = None                                           # release before allocating the next size
            stream = pkg.synth_stream(plan['first'], n_frames, seed=1337)   # on the device, outside the timing
            buffers = ((plan['first'], n_frames), stream, pkg.StreamWorkspace(n_frames, stream.device))
        _, stream, ws = buffers
        net = make_net(model)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

        def step(events=None):
            if events:
                events[0].record()
            pkg.mfcc_delta_stream(stream, n_frames, args.slice, out=ws.feat)
            if events:
                events[1].record()
            net.score(ws.feat, window_stride=24, n_windows=n_windows, out=ws.p, decisions=ws.decision)
            if events:
                events[2].record()

        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            step(ev[k])
        barrier()
        elapsed = time.perf_counter() - t0
        if dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            tot = torch.tensor([plan['n_own']], dtype=torch.int64, device=red_dev)
            dist.all_reduce(tot)
            total_frames = int(tot.item())
        else:
            total_frames = plan['n_own']
        if rank != 0:
            return None, buffers

        # events are recorded on the stream the kernels are launched on (torch's current stream)
        fe_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        nn_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        assert bool(torch.isfinite(ws.p).all())
        hours_per_step = total_frames / FRAMES_PER_HOUR
        value = hours_per_step * steps / elapsed
        flop_alg, flop_torch = FLOP_PER_WINDOW[model]
        achieved = flop_alg * n_windows / (nn_ms * 1e-3) / 1e12
        # fp32 MFMA (v_mfma_f32_32x32x2_f32) issues at the fp32 vector FMA rate and does not co-execute with
        # vector instructions, so matrix and vector fp32 work share the one 157.3 TFLOP/s peak.
        kernel = model_kernel(model, current_mode[0])
        nn_pmc = pmc_per_unit(prof, kernel)
        model_roof = {
            'bound': 'mfma', 'kernel': kernel, 'model': model,
            'matrix_mode': current_mode[0] if kernel != MODEL_KERNEL[model] else 'f32',
            'achieved': achieved, 'peak': model_peak(model, current_mode[0]), 'unit': 'TFLOP/s',
            'frac': achieved / model_peak(model, current_mode[0]), 'frac_of_fp32_peak': achieved / PEAK_FP32_TFLOPS,
            'traffic': nn_pmc['bytes_per_unit'] * n_windows if nn_pmc and 'bytes_per_unit' in nn_pmc else None,
            'traffic_from_profile': nn_pmc, 'ms_per_launch': nn_ms,
            'algorithmic_bytes': (F * 4 + RESULT_BYTES_PER_WINDOW) * n_windows,
            'flop_per_window': flop_alg, 'flop_per_window_pytorch': flop_torch, 'windows_per_launch': n_windows,
            'note': 'achieved = algorithmic float32 flops / time, with the per-row work (input projection / DenseNet '
                    'stem) counted once per feature row as stride-1 windows need it; flop_per_window_pytorch counts '
                    'it per (30,24) window as a batch framework does.  peak: 157.3 TF (fp32 MFMA = fp32 vector rate) '
                    'for float32-MFMA kernels; in the split matrix modes (bf16x9 / bf16x8 / bf16x6 / f16x6 / f16x4) every '
                    'float32 product of the recurrence and the 26-row head costs 9 / 8 / 6 / 6 / 4 products of the 16-bit '
                    'matrix pipe, so the peak for that work is 2500 TF / that count, taken beside the float32 pipe for the rest'}
        fe_gbs = FRONTEND_BYTES_PER_FRAME * n_frames / (fe_ms * 1e-3) / 1e9
        fe_tf = FRONTEND_FLOP_PER_FRAME * n_frames / (fe_ms * 1e-3) / 1e12
        fe_pmc = pmc_per_unit(prof, 'mfcc_kernel')
        # The front end is compute-bound (61 flop per byte): it is priced against the float32 peak (matrix = vector rate),
        # with its HBM rate beside it.
        fe_roof = {
            'bound': 'mfma', 'kernel': 'mfcc_kernel', 'achieved': fe_tf, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
            'frac': fe_tf / PEAK_FP32_TFLOPS, 'hbm_GBs': fe_gbs, 'hbm_frac': fe_gbs / PEAK_HBM_GBS,
            'traffic': fe_pmc['bytes_per_unit'] * n_frames if fe_pmc and 'bytes_per_unit' in fe_pmc else None,
            'traffic_from_profile': fe_pmc, 'ms_per_launch': fe_ms,
            'algorithmic_bytes': FRONTEND_BYTES_PER_FRAME * n_frames, 'bytes_per_frame': FRONTEND_BYTES_PER_FRAME,
            'frames_per_launch': n_frames, 'flop_per_frame': FRONTEND_FLOP_PER_FRAME, 'fp32_TFLOPs': fe_tf,
            'fp32_frac': fe_tf / PEAK_FP32_TFLOPS,
            'note': 'the kernel is not HBM-bound (61 flop per byte); its FFT is add-dominated (one flop per '
                    'instruction slot), so the FMA peak is out of reach by construction and the second yardstick '
                    'is vector-instruction issue: valu_issue'}
        if fe_pmc and 'valu_insts_per_unit' in fe_pmc:
            # the time the chip needs just to ISSUE the kernel's vector instructions at 2.4 GHz
            issue_ms = fe_pmc['valu_insts_per_unit'] * n_frames * VALU_ISSUE_CYCLES / (SIMDS * 2.4e9) * 1e3
            # the same at the rate a SIMD sustains on independent v_add_f32 / v_fmac_f32 with the kernel's three
            # waves per SIMD (tools/probes/valu_issue_probe.hip, profiles/r02_valu_issue_probe.txt)
            measured_ms = fe_pmc['valu_insts_per_unit'] * n_frames * VALU_ISSUE_NS_MEASURED / SIMDS * 1e-6
            fe_roof['valu_issue'] = {'valu_wave_insts_per_frame': fe_pmc['valu_insts_per_unit'],
                                     'cycles_per_inst': VALU_ISSUE_CYCLES, 'bound_ms': issue_ms,
                                     'frac': issue_ms / fe_ms,
                                     'measured_ns_per_inst': VALU_ISSUE_NS_MEASURED, 'measured_bound_ms': measured_ms,
                                     'measured_frac': measured_ms / fe_ms}
        rec = {
            'value': value, 'unit': 'audio-hours/s', 'steps': steps, 'warmup': warmup,
            'ms_per_step': elapsed / steps * 1e3,
            'config': {'workload': '%s, %.0f h synthetic 16 kHz int16 audio per GPU, 12 MFCC + 12 delta, '
                                   '%d-frame slices, stride-1 900 ms windows' % (model, hours, args.slice),
                       'frames_per_gpu': n_frames, 'windows_per_gpu': n_windows, 'parallelism': 'shards x%d' % world},
            'x_realtime_per_gpu': value * 3600.0 / world,
            # the dominant kernel: the longer of the two; when they are within 5 % of each other (less than the
            # spread between devices of one pool) the one that carries more of the step's arithmetic
            'roofline': (model_roof if flop_alg * n_windows >= FRONTEND_FLOP_PER_FRAME * n_frames else fe_roof)
                        if abs(nn_ms - fe_ms) <= 0.05 * max(nn_ms, fe_ms) else (model_roof if nn_ms >= fe_ms else fe_roof),
            'roofline_choice': 'model %.2f ms, front end %.2f ms per launch: %s' % (
                nn_ms, fe_ms, 'within 5 %, the kernel with more algorithmic flops is reported'
                if abs(nn_ms - fe_ms) <= 0.05 * max(nn_ms, fe_ms) else 'the longer kernel is reported'),
            'kernels': {'model': model_roof, 'frontend': fe_roof},
        }
        return rec, buffers

    head, buffers = run_workload(args.model, args.hours, args.steps, args.warmup)
    out = None
    if rank == 0:
        out = {'metric': 'audio-hours processed/sec (MFCC+GRU-30k VAD)' if args.model == 'gru_large'
                         else 'audio-hours processed/sec (MFCC+%s VAD)' % args.model,
               'value': head['value'], 'unit': head['unit'], 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': head['ms_per_step'], 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': head['config'], 'x_realtime_per_gpu': head['x_realtime_per_gpu'],
               'roofline': head['roofline'], 'roofline_choice': head['roofline_choice'], 'kernels': head['kernels'],
               'matrix_mode': args.matrix_mode,
               'dtype_note': 'float32 MFMA (k-ordered FMA chain)' if args.matrix_mode == 'f32' else
                             'f32 operands split exactly into 3 %s, %s of 9 partial products on the 16-bit MFMA, f32 '
                             'accumulation' % ('f16' if args.matrix_mode.startswith('f16') else 'bf16', args.matrix_mode[-1]),
               'dist_backend': (backend if dist else None)}

    if not args.no_configs and args.model == 'gru_large' and args.matrix_mode == DEFAULT_MODE:
        # The other configurations of the measurement plan, 3 steps each on the buffers of the headline run:
        # LSTM-small, DenseNet small and large, the enlarged GRU (hidden 90) and NickNet.  On request the enlarged
        # GRU also runs at its named size, 10 000 h over 8 GPUs = 1 250 h (144 GB of int16) per GPU.
        extra = {}
        for model in ('lstm_small', 'densenet', 'densenet_large', 'gru_h90', 'nicknet'):
            rec, buffers = run_workload(model, args.hours, 3, 1, buffers)
            extra[model] = rec
        # Opt-in (BENCH_BIG_WORLD=<world size>, BENCH_BIG_HOURS): the line the driver parses must not depend on a
        # 174 GB-per-GPU leg that no multi-rank run has exercised; one such shard runs through this branch on one
        # GPU with BENCH_BIG_WORLD=1 (profiles/r03_d_bench_1250h_shard.log).
        big_hours = float(os.environ.get('BENCH_BIG_HOURS', '1250'))
        if world == int(os.environ.get('BENCH_BIG_WORLD', '0')):
            # needs 144 GB of audio + 14 GB of features + 16 GB of workspace per GPU; every rank must take the
            # same decision, so the ranks agree on it (and on whether their allocations succeeded) first
            buffers = None
            torch.cuda.empty_cache()

            def all_agree(ok):
                if not dist:                                     # single-process rehearsal (BENCH_BIG_WORLD=1)
                    return bool(ok)
                flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=red_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return bool(flag.item())

            if all_agree(torch.cuda.mem_get_info()[0] > big_hours * 0.16e9):
                plan = shard_plan(int(round(big_hours * FRAMES_PER_HOUR)) * world, world, rank, args.slice)
                try:
                    stream = pkg.synth_stream(plan['first'], plan['n_load'], seed=1337)
                    big = ((plan['first'], plan['n_load']), stream, pkg.StreamWorkspace(plan['n_load'], stream.device))
                    torch.cuda.synchronize()
                except (RuntimeError, MemoryError):
                    big = None
                if all_agree(big is not None):
                    rec, big = run_workload('gru_h90', big_hours, 2, 1, big)
                    extra['gru_h90_%dh' % round(big_hours * world)] = rec
                big = None
        if rank == 0:
            out['configs'] = extra

    if rank == 0 and world == 1 and args.matrix_mode == DEFAULT_MODE and \
            args.model in ('gru_large', 'gru_small', 'lstm_large', 'lstm_small'):
        # Not part of `value`: the same step in the other matrix modes of the stride-1 GRU / LSTM kernels
        # (vad_set_matrix_mode) -- the float32 MFMA (a k-ordered FMA chain), bf16x9 and bf16x6 -- and how far the
        # posteriors of each mode are from the float32-MFMA ones over every window of the workload.
        alt, ps = {}, {}
        for mm in ('f32', 'f16x6', 'f16x4', 'bf16x9', 'bf16x8', 'bf16x6'):
            pkg.set_matrix_mode(mm)
            current_mode[0] = mm
            rec, buffers = run_workload(args.model, args.hours, 3, 1, buffers)
            ps[mm] = buffers[2].p.clone()
            alt[mm] = {'value': rec['value'], 'unit': 'audio-hours/s', 'ms_per_step': rec['ms_per_step'],
                       'model_kernel_ms': rec['kernels']['model']['ms_per_launch'],
                       'model_kernel_frac': rec['kernels']['model']['frac']}
        for mm in alt:
            alt[mm]['max_abs_posterior_diff_vs_f32'] = float((ps[mm] - ps['f32']).abs().max())
        pkg.set_matrix_mode(args.matrix_mode)
        current_mode[0] = args.matrix_mode
        out['matrix_modes'] = alt
        del ps
    if rank == 0:
        if world == 1 and not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline(args.model)
        out['detail'] = write_detail(out)
        print(compact_line(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
