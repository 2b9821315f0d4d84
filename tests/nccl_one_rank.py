"""Helper of test_gpu_sharding_nccl.py, run as a child process: one rank, backend nccl (= RCCL), on cuda:0.
Scores a small stream through the product path, then takes the posteriors through every collective the
multi-rank path uses: barrier, MAX / SUM all_reduce on device tensors, all_gather_posteriors."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import voice_activity_detection_amd as pkg
    from voice_activity_detection_amd.sharding import all_gather_posteriors, gather_posteriors, shard_plan

    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(s.getsockname()[1]), RANK='0', WORLD_SIZE='1')
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    assert dist.get_backend() == 'nccl'
    n_frames = 166 * 4 + 50
    plan = shard_plan(n_frames, 1, 0, 166)
    assert plan['n_load'] == n_frames and plan['n_windows'] == n_frames - 29
    stream = pkg.synth_stream(plan['first'], plan['n_load'], seed=1337)
    torch.manual_seed(1337)
    net = pkg.Net(large=True, lstm=False)
    _, p, dec = pkg.vad_stream(net, stream, n_frames, 166)
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([plan['n_own']], dtype=torch.int64, device='cuda')
    dist.all_reduce(n)
    assert float(t.item()) == 1.25 and int(n.item()) == n_frames
    everywhere = all_gather_posteriors(p, 1, dist)           # device tensors through RCCL
    assert everywhere.is_cuda and everywhere.data_ptr() != p.data_ptr()
    assert torch.equal(everywhere, p)
    assert torch.equal(gather_posteriors(p, 1, dist), p.cpu())
    d_all = all_gather_posteriors(dec, 1, dist)              # the uint8 decisions take the same route
    assert torch.equal(d_all, dec)
    dist.barrier()
    dist.destroy_process_group()
    print('nccl one-rank OK: %d windows gathered on %s' % (everywhere.shape[0], everywhere.device))


if __name__ == '__main__':
    main()
