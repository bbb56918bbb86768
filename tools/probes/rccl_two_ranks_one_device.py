"""Probe: does this RCCL accept two ranks on ONE device (torch.distributed backend "nccl" and the C ABI's own communicator)? Prints what happened; never hangs
(each attempt runs in spawned processes under a timeout)."""
import os, socket, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def job(rank, world, prt, mode, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(prt)
    torch.cuda.set_device(0)
    try:
        if mode == "torch-nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world)
            t = torch.full((1024,), float(rank + 1), device="cuda:0")
            dist.all_reduce(t)
            torch.cuda.synchronize()
            ret[rank] = f"ok sum={float(t[0])}"
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            import studiogan_amd
            from studiogan_amd import comm
            nc = comm.enable(dist.group.WORLD, device=torch.device("cpu"))
            t = torch.full((1024,), float(rank + 1), device="cuda:0")
            nc.allreduce_(t)
            torch.cuda.synchronize()
            ret[rank] = f"ok sum={float(t[0])}"
    except Exception as e:  # noqa: BLE001
        ret[rank] = f"{type(e).__name__}: {str(e)[:300]}"
    finally:
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


if __name__ == "__main__":
    for mode in ("torch-nccl", "native"):
        for env in ({}, {"NCCL_IGNORE_DUPLICATE_GPU": "1", "RCCL_IGNORE_DUPLICATE_GPU": "1"}):
            os.environ.update(env)
            mgr = mp.Manager(); ret = mgr.dict()
            ctx = mp.spawn(job, args=(2, port(), mode, ret), nprocs=2, join=False)
            t0 = time.time()
            while not ctx.join(timeout=1):
                if time.time() - t0 > 60:
                    for p in ctx.processes:
                        p.kill()
                    ret["timeout"] = True
                    break
            print(mode, env, dict(ret), flush=True)
