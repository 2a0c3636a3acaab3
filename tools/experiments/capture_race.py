"""Does the process group's watchdog, polling the events of EAGER collectives, collide with a stream capture that pulls the same RCCL
stream in?  (world size 1 over nccl on one GPU.)  usage: capture_race.py shared|separate [rounds]"""
import os, sys, time
import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else 'shared'
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
gpg = None
if mode == 'separate':
    gpg = dist.new_group(backend='nccl')
    w = torch.ones(8, device='cuda'); dist.all_reduce(w, group=gpg); torch.cuda.synchronize(); time.sleep(0.5)
s = torch.cuda.Stream()
x = torch.ones(1024, device='cuda', dtype=torch.float64)
y = torch.ones(1024, device='cuda', dtype=torch.float64)
bad = 0
t0 = time.time()
with torch.cuda.stream(s):
    for r in range(rounds):
        for _ in range(15):
            dist.all_reduce(x)                      # eager works: stay in the watchdog's list until its next poll
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                for _ in range(12):
                    y.mul_(1.0)
                    dist.all_reduce(y, group=gpg)
            g.replay()
        except Exception as e:
            bad += 1
            print('round', r, type(e).__name__, str(e).splitlines()[0]); sys.stdout.flush()
            break
        time.sleep(0.003 * (r % 40))                # walk the capture across the watchdog's polling phase
torch.cuda.synchronize()
print(mode, 'rounds', r + 1, 'failures', bad, '%.1f s' % (time.time() - t0))
dist.destroy_process_group()
