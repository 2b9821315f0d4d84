#!/usr/bin/env python3
"""Throughput of the VAD hot path on MI355X: audio-hours processed per second.

One "step" is one pass of the whole path over one batch of synthetic audio that
is already resident in HBM: int16 frames -> MFCC+delta (166-frame slices, as
process_test_data cuts them) -> stride-1 30-frame windows -> GRU-30k forward ->
posterior + decision per window, all left in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--hours H] [--model gru_large]

For N > 1 the driver launches one rank per GPU through torch.distributed.run;
every rank scores its own contiguous shard of the stream (weak scaling, no
data-path collective); the timed region is bracketed by a barrier and the
slowest rank's time is used.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_HOUR = 120000          # 3600 s / 0.03 s
# Algorithmic work per unit, PyTorch-equivalent operation counts.
FLOP_PER_WINDOW = {               # per 30 x 24 window
    'gru_large': 30 * 24 * 90 * 2 + 30 * 30 * 90 * 2 + 30 * 30 * 12 + 900 * 26 * 2 + 26 * 2 * 2,
    'gru_small': 30 * 24 * 90 * 2 + 30 * 30 * 90 * 2 + 30 * 30 * 12 + 900 * 2 * 2,
    'lstm_large': 30 * 24 * 120 * 2 + 30 * 30 * 120 * 2 + 30 * 30 * 15 + 900 * 26 * 2 + 26 * 2 * 2,
    'lstm_small': 30 * 24 * 120 * 2 + 30 * 30 * 120 * 2 + 30 * 30 * 15 + 900 * 2 * 2,
    'gru_h90': 30 * 24 * 270 * 2 + 30 * 90 * 270 * 2 + 30 * 90 * 12 + 2700 * 26 * 2 + 26 * 2 * 2,   # enlarged GRU, ~100k params
    'densenet': 103_000,
    'densenet_large': 262_000,
}
FRONTEND_BYTES_PER_FRAME = 480 * 2 + 24 * 4          # int16 in, 24 floats out
FRONTEND_FLOP_PER_FRAME = 64_000                       # 2048-pt rfft + power + mel + DCT + delta
PEAK_FP32_TFLOPS = 157.3                               # dense fp32 matrix = fp32 vector peak, MI355X
PEAK_HBM_GBS = 8000.0
# What the stride-1 GRU/LSTM kernels execute per window (padded 32-row tiles, projection hoisted to once per row):
# 30 steps x (gate tiles + head tile) x 15 MFMAs x 4096 flop / 32 windows + the row's projection + gate math.
EXECUTED_FLOP_PER_WINDOW = {
    'gru_large': 30 * 4 * 15 * 4096 // 32 + 3 * 12 * 4096 // 32 + 30 * 32 * 13,
    'lstm_large': 30 * 5 * 15 * 4096 // 32 + 4 * 12 * 4096 // 32 + 30 * 32 * 18,
}
# HBM bytes per frame from rocprofv3 --pmc passes over this same command at 100 h (12,000,059 frames per launch;
# profiles/r01_h_bench_pmc_100h.txt): 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of coalesced streaming
# reads at 64 bytes; unscaled, mfcc_kernel's figure would be 0.63x the input it must read) + WRITE_SIZE, KiB -> bytes.  Counters cannot be read from inside this process, so the
# per-launch figure is this measured per-frame constant times the frames of the launch.
_KIB_PER_LAUNCH = {'frontend': 2 * (7073810 + 702013) + 750000 + 755859,      # mfcc_kernel + delta_kernel
                   'gru_large': 2 * 633401 + 105469}                            # rnn_stream_kernel<GRU, large>
TRAFFIC_BYTES_PER_FRAME = {k: v * 1024 / 12000059 for k, v in _KIB_PER_LAUNCH.items()}


def make_net(kind):
    """The product model with seeded random weights (no checkpoints ship): torch's default initialisation of
    the modules under the reference's seed (core/train.py:56), BatchNorm statistics perturbed away from the
    identity.  Nothing from oracle/ is involved in the measured path."""
    import voice_activity_detection_amd as pkg
    torch.manual_seed(1337)
    if kind == 'gru_h90':                              # no class in the reference: GRU(24, 90) + the same heads
        return pkg.Net(large=True, lstm=False, hidden=90)
    if kind.startswith('densenet'):
        net = pkg.DenseNet(large=kind.endswith('large'))
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
    from oracle import models_ref
    net = make_net(kind)
    ref = models_ref.make_model('gru_large', hidden=90) if kind == 'gru_h90' else models_ref.make_model(kind)
    ref.load_state_dict(net.state_dict())
    return net, ref.eval()


def cpu_baseline(kind, budget_s=12.0):
    """The CPU oracle (float64 numpy MFCC + torch CPU model, batch 2048) on a bounded sample: once with torch's
    default thread count, once (shorter) with a single thread."""
    from oracle import pipeline_ref, synth_ref
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

    threads = torch.get_num_threads()
    reps, dt = leg(budget_s)
    torch.set_num_threads(1)
    reps1, dt1 = leg(budget_s / 3)
    torch.set_num_threads(threads)
    return {'value': reps * 2000 / FRAMES_PER_HOUR / dt, 'unit': 'audio-hours/s', 'cores': threads, 'kind': 'port',
            'single_thread_value': reps1 * 2000 / FRAMES_PER_HOUR / dt1,
            'sample': '%d x 1 min synthetic audio (2000 frames, 1971 windows each), float64 numpy/scipy MFCC+delta '
                      'in 166-frame slices + torch CPU %s, batch 2048; %.1f s of CPU work (+ %.1f s single-threaded), '
                      'host has %d CPUs' % (reps, kind, dt, dt1, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--hours', type=float, default=100.0, help='audio hours per GPU per step')
    ap.add_argument('--model', default='gru_large', choices=sorted(FLOP_PER_WINDOW))
    ap.add_argument('--slice', type=int, default=166, help='frames per MFCC slice (reference test path: 166)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--matrix-mode', default='f32', choices=['f32', 'bf16x6', 'bf16x9'],
                    help='how the stride-1 GRU/LSTM kernels form float32 products (default: float32 MFMA)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
        args.gpus = world
    # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share a GPU when rehearsing the
    # multi-rank path on a one-GPU box (timing is then meaningless, the control flow is not)
    backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
    local = local % max(torch.cuda.device_count(), 1) if backend == 'gloo' else local
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
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
    net = make_net(args.model)

    frames_per_gpu = int(round(args.hours * FRAMES_PER_HOUR))
    # contiguous slice-aligned shard of one stream; the rank re-creates its own lead-in and halo,
    # and scores exactly the windows that start inside its shard (no inter-rank traffic)
    plan = shard_plan(frames_per_gpu * world, world, rank, args.slice)
    n_frames, n_windows = plan['n_load'], plan['n_windows']
    stream = pkg.synth_stream(plan['first'], n_frames, seed=1337)   # generated on the device, outside the timing
    ws = pkg.StreamWorkspace(n_frames, stream.device)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(events=None):
        if events:
            events[0].record()
        pkg.mfcc_delta_stream(stream, n_frames, args.slice, out=ws.feat)
        if events:
            events[1].record()
        net.score(ws.feat, window_stride=24, n_windows=n_windows, out=ws.p, decisions=ws.decision)
        if events:
            events[2].record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
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

    if rank == 0:
        fe_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        nn_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        assert bool(torch.isfinite(ws.p).all())
        hours_per_step = total_frames / FRAMES_PER_HOUR
        value = hours_per_step * args.steps / elapsed
        flop = FLOP_PER_WINDOW[args.model] * n_windows
        achieved = flop / (nn_ms * 1e-3) / 1e12
        fe_tf = FRONTEND_FLOP_PER_FRAME * n_frames / (fe_ms * 1e-3) / 1e12
        fe_gbs = FRONTEND_BYTES_PER_FRAME * n_frames / (fe_ms * 1e-3) / 1e9
        # Both kernels are bound by the SIMDs' fp32 FMA lanes: v_mfma_f32_32x32x2_f32 runs at the fp32 vector
        # rate and never co-executes with vector instructions (SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/),
        # so fp32 MFMA and fp32 VALU share one 157.3 TFLOP/s peak.
        nn_traffic = TRAFFIC_BYTES_PER_FRAME.get(args.model)
        model_roof = {'bound': 'mfma', 'kernel': 'model kernel (%s): rnn_stream_kernel / densenet_kernel' % args.model,
                      'achieved': achieved, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                      'frac': achieved / PEAK_FP32_TFLOPS,
                      'traffic': None if nn_traffic is None else nn_traffic * n_windows, 'ms_per_launch': nn_ms,
                      'algorithmic_bytes': (24 * 4 + 9) * n_windows,
                      'executed_flop_per_window': EXECUTED_FLOP_PER_WINDOW.get(args.model),
                      'frac_executed': None if args.model not in EXECUTED_FLOP_PER_WINDOW else
                      EXECUTED_FLOP_PER_WINDOW[args.model] * n_windows / (nn_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                      'flop_per_window': FLOP_PER_WINDOW[args.model], 'windows_per_launch': n_windows,
                      'note': 'achieved = algorithmic (PyTorch-equivalent) flops / time; the stride-1 GRU/LSTM kernel '
                              'projects each feature row once instead of once per window, so it executes ~0.69x of them'}
        fe_roof = {'bound': 'mfma', 'kernel': 'mfcc_kernel (+ delta_kernel)', 'achieved': fe_tf, 'peak': PEAK_FP32_TFLOPS,
                   'unit': 'TFLOP/s', 'frac': fe_tf / PEAK_FP32_TFLOPS,
                   'traffic': TRAFFIC_BYTES_PER_FRAME['frontend'] * n_frames, 'ms_per_launch': fe_ms,
                   'algorithmic_bytes': FRONTEND_BYTES_PER_FRAME * n_frames,
                   'flop_per_frame': FRONTEND_FLOP_PER_FRAME, 'frames_per_launch': n_frames,
                   'pipe': 'fp32 vector ALU (same FMA lanes as fp32 MFMA)',
                   'hbm_GBs': fe_gbs, 'hbm_frac': fe_gbs / PEAK_HBM_GBS, 'bytes_per_frame': FRONTEND_BYTES_PER_FRAME}
        out = {
            'metric': 'audio-hours processed/sec (MFCC+GRU-30k VAD)' if args.model == 'gru_large'
                      else 'audio-hours processed/sec (MFCC+%s VAD)' % args.model, 'value': value, 'unit': 'audio-hours/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s, %.0f h synthetic 16 kHz int16 audio per GPU, 12 MFCC + 12 delta, '
                                   '%d-frame slices, stride-1 900 ms windows' % (args.model, args.hours, args.slice),
                       'frames_per_gpu': n_frames, 'windows_per_gpu': n_windows, 'parallelism': 'shards x%d' % world},
            'x_realtime_per_gpu': value * 3600.0 / world,
            'roofline': model_roof if nn_ms >= fe_ms else fe_roof,          # the longer of the two kernels
            'kernels': {'model': model_roof, 'frontend': fe_roof},
        }
        if world == 1 and args.matrix_mode == 'f32' and args.model in ('gru_large', 'gru_small', 'lstm_large', 'lstm_small'):
            # Not part of `value`: the same step with the recurrent and head products on the bf16 matrix pipe
            # (float32 operands split exactly into three bf16; vad_set_matrix_mode), and how far its posteriors
            # are from the float32-MFMA ones over every window of the workload.
            p_f32 = ws.p.clone()
            alt = {}
            for mm in ('bf16x9', 'bf16x6'):
                pkg.set_matrix_mode(mm)
                step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 3
                alt[mm] = {'value': hours_per_step / dt, 'unit': 'audio-hours/s', 'ms_per_step': dt * 1e3,
                           'max_abs_posterior_diff_vs_f32': float((ws.p - p_f32).abs().max())}
            pkg.set_matrix_mode('f32')
            out['alt_matrix_modes'] = alt
            del p_f32
        if world == 1 and not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline(args.model)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
