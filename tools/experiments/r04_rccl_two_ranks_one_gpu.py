"""Does this RCCL accept two ranks on ONE device?  (If it did, the library's N > 1 path could be exercised on a one-GPU box.)
Two processes, both on cuda:0, join one communicator through the library (nw_comm_init) and all-reduce four numbers.
usage: python tools/experiments/r04_rccl_two_ranks_one_gpu.py"""
import os, sys, socket, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(rank, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=2)
    try:
        from ch_shrinkwrap_amd import parallel
        from ch_shrinkwrap_amd.mesh_conj_grad import NativeContext
        native = NativeContext(0)
        try:
            comm = parallel.NativeComm(native, dist)
            a = comm.all_reduce_host(np.array([1.0 + rank, 2.0, 3.0, 4.0], np.float64))
            q.put((rank, 'ok', a.tolist()))
            comm.close()
        except Exception as e:
            q.put((rank, 'refused', str(e)[:300]))
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    for _ in range(2):
        try:
            print(q.get(timeout=60))
        except Exception:
            print('no answer within 60 s')
    for p in ps:
        p.join(timeout=10)
        if p.is_alive():
            p.terminate()
