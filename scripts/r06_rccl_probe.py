import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
from pydreamer_amd import hip as H
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29544')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
x = torch.randn(23_000_000, device='cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    th=time.perf_counter()-t
    torch.cuda.synchronize(); return 1e3*th/n, 1e3*(time.perf_counter()-t)/n
print('torch all_reduce 92 MB, 1 rank: host %.3f ms/call, total %.3f ms/call' % timeit(lambda: dist.all_reduce(x)))
idb=(ctypes.c_char*128)(); H.call('dm_rccl_unique_id', idb); comm=ctypes.c_void_p()
H.call('dm_rccl_comm_init', ctypes.byref(comm), 1, ctypes.c_char_p(bytes(idb)), 0)
print('native dm_allreduce_grads 92 MB: host %.3f ms/call, total %.3f ms/call' % timeit(lambda: H.call('dm_allreduce_grads', H.fptr(x), x.numel(), comm, H.stream())))
y = torch.randn(1000, device='cuda')
print('native 4 KB: host %.3f ms/call, total %.3f' % timeit(lambda: H.call('dm_allreduce_grads', H.fptr(y), y.numel(), comm, H.stream())))
print('torch 4 KB: host %.3f ms/call, total %.3f' % timeit(lambda: dist.all_reduce(y)))
s2=torch.cuda.Stream()
with torch.cuda.stream(s2):
    print('native 92 MB on a side stream: host %.3f, total %.3f' % timeit(lambda: H.call('dm_allreduce_grads', H.fptr(x), x.numel(), comm, H.stream())))
