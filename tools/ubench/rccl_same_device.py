#!/usr/bin/env python3
"""Does RCCL accept two ranks on ONE device on this box?  (It decides how the multi-process strip tests can run on a 1-GPU lease.)
    python tools/ubench/rccl_same_device.py          -> starts two children, prints what happened"""
import os
import subprocess
import sys


def child(rank):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=2)
    t = torch.ones(4, device="cuda:0") * (rank + 1)
    try:
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print("rank %d: all_reduce on a shared device -> %s" % (rank, t.tolist()))
    except Exception as ex:  # noqa: BLE001
        print("rank %d: FAILED: %s" % (rank, str(ex).splitlines()[0][:300]))
    sys.stdout.flush()
    os._exit(0)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r)], env=env) for r in range(2)]
    for p in ps:
        try:
            p.wait(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            print("timeout")
