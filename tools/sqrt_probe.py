import ctypes as C, torch, sys
sys.path.insert(0,'/root/repo')
from multiagent_rl_amd import _lib
lib=_lib.load()
stream=C.c_void_p(torch.cuda.current_stream().cuda_stream)
sig=torch.arange(1<<23,dtype=torch.int32,device='cuda')
o4,o11,o12=[torch.empty(1<<23,device='cuda') for _ in range(3)]
bad11=bad12=0
for e in range(-90,90):
    x=(sig|((e+127)<<23)).view(torch.float32)
    for fn,o in ((4,o4),(11,o11),(12,o12)):
        assert lib.pw_debug_math(fn,C.c_void_p(x.data_ptr()),C.c_float(1.0),C.c_void_p(o.data_ptr()),x.numel(),stream)==0
    n11=int((o4.view(torch.int32)!=o11.view(torch.int32)).sum()); n12=int((o4.view(torch.int32)!=o12.view(torch.int32)).sum())
    bad11+=n11; bad12+=n12
    if e in (-90,-1,0,1,89) or n11: print('exp',e,'core1 mismatches',n11,'core mismatches',n12)
print('total mismatches: core1',bad11,'core',bad12)
