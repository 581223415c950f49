"""Latency of the gradient exchange's collective on a process group of ONE rank over RCCL (the only RCCL group a
one-GPU box can form): torch.distributed.all_reduce of the cfg2 / cfg4 gradient vector (22,913 floats = 92 KB) and of
the cfg3 one (354,201 floats = 1.42 MB). With one rank there is no wire: what is measured is the fixed cost every step
pays whatever the world size - the enqueue on the host and the RCCL kernel's launch + completion on the device.
Prints one JSON line. Usage (GPU box): python tools/allreduce_microbench.py"""
import json
import os
import socket
import time

import torch
import torch.distributed as dist

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
if 'MASTER_PORT' not in os.environ:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        os.environ['MASTER_PORT'] = str(s.getsockname()[1])
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
out = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'sizes': {}}
for name, n in (('cfg2_cfg4_22913_floats', 22913), ('cfg3_354201_floats', 354201)):
    buf = torch.randn(n, device='cuda')
    for _ in range(20):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    # (a) one collective at a time, host waits for it: launch + completion latency
    lat = []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.all_reduce(buf)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    lat.sort()
    # (b) device time of the collective itself (events around it on its stream), back to back
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        dist.all_reduce(buf)
    e1.record()
    torch.cuda.synchronize()
    # (c) host enqueue cost alone (async_op, no wait)
    t0 = time.perf_counter()
    works = [dist.all_reduce(buf, async_op=True) for _ in range(200)]
    enq = (time.perf_counter() - t0) / 200
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    out['sizes'][name] = {'bytes': 4 * n, 'sync_latency_us_median': round(lat[100] * 1e6, 1),
                          'sync_latency_us_p10_p90': [round(lat[20] * 1e6, 1), round(lat[180] * 1e6, 1)],
                          'device_us_back_to_back': round(e0.elapsed_time(e1) * 1e3 / 200, 1),
                          'host_enqueue_us': round(enq * 1e6, 1)}
print(json.dumps(out))
dist.destroy_process_group()
