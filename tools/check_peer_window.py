#!/usr/bin/env python3
"""Two processes, one GPU: rank 0 exports a device allocation (sfx_peer_export), rank 1 maps it and copies into it (sfx_peer_copy)."""
import os
import sys
import traceback

import numpy as np
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rank(r, queue_to, queue_from):
    try:
        from shaderflow_amd import _native as N
        context = N.Context(0, None)
        n = 1 << 20
        if r == 0:
            pointer = context.alloc(n)
            queue_to.put(context.peer_export(pointer))
            assert queue_from.get(timeout=60) == "copied"
            got = context.read(pointer, n)
            print("rank 0 sees", got[:4], got[-4:], flush=True)
            assert (got == 7).all()
            queue_to.put("done")
            context.free(pointer)
        else:
            handle = queue_from.get(timeout=60)
            print("rank 1 got handle", len(handle), flush=True)
            window = context.peer_open(handle)
            print("rank 1 opened", hex(window), flush=True)
            import torch
            source = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            context.peer_copy(window, source.data_ptr(), n, 0)
            context.peer_flush()
            print("rank 1 copied", flush=True)
            queue_to.put("copied")
            assert queue_from.get(timeout=60) == "done"
            context.peer_close(window)
    except Exception:
        traceback.print_exc()
        raise


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    a, b = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=rank, args=(0, a, b)), ctx.Process(target=rank, args=(1, b, a))]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    print("exit codes", [p.exitcode for p in procs])
