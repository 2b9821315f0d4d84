"""The RCCL side of the multi-rank path on a one-GPU box: a single-rank nccl process group, device tensors
through barrier / all_reduce / all_gather_into_tensor (sharding.all_gather_posteriors).  The partitioning
itself is covered with two gloo ranks on the CPU (test_sharding_gloo.py); this checks that the calls the
GPU ranks make work on device memory with the backend the GPU ranks use."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_one_rank_nccl_group_gathers_device_posteriors(vad):
    # a child process: the process group (and RCCL's own streams) stay out of the test process
    proc = subprocess.run([sys.executable, os.path.join(HERE, 'nccl_one_rank.py')], capture_output=True, text=True,
                          timeout=300)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    assert 'nccl one-rank OK' in proc.stdout
