#!/usr/bin/env python3
"""Smoke-test bench.py's N>1 code path on a ONE-GPU box: every rank uses cuda:0 and the collectives run over gloo with
CPU staging (RCCL refuses two ranks on one device).  Exercises exactly the sharded code bench.py runs under RCCL:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        tools/bench_multi_smoke.py --gpus 2 --steps 3 --warmup 1 --config 256
"""
import os, sys
import torch
import torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)

_init = dist.init_process_group
def init(backend=None, **kw):
    kw.pop("device_id", None)
    return _init("gloo", **kw)
dist.init_process_group = init
torch.cuda.set_device = lambda d: None            # everyone on cuda:0

def staged(fn, inplace_arg=0):
    def wrap(t, *a, **kw):
        c = t.cpu()
        fn(c, *a, **kw)
        t.copy_(c)
    return wrap
dist.broadcast = staged(dist.broadcast)
dist.all_reduce = staged(dist.all_reduce)
dist.reduce = staged(dist.reduce)
_batch = dist.batch_isend_irecv
class _Done:
    def wait(self): pass
def batch(ops):
    cpu_ops, backs = [], []
    for op in ops:
        c = op.tensor.cpu()
        cpu_ops.append(dist.P2POp(op.op, c, op.peer, op.group))
        if op.op is dist.irecv: backs.append((op.tensor, c))
    for r in _batch(cpu_ops): r.wait()
    for t, c in backs: t.copy_(c)
    return [_Done()]
dist.batch_isend_irecv = batch
_barrier = dist.barrier
dist.barrier = lambda *a, **k: _barrier()

import bench
bench.main()
