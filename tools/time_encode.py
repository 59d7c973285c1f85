import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from ldpc_erasure_codes_amd import api, codes
ctx = api.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
h = ctx.load_builtin_code(1, 2040); n,k,_ = ctx.code_info(h)
F,S = 4096,1024
src = torch.empty((F,k,S),dtype=torch.uint8,device="cuda"); ctx.synth_source(1,0,F,k,S,src)
for mode in ("gather","scatter"):
    os.environ["LDPC_AMD_APPLY"]=mode
    cw = ctx.encode(h, src); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(5): cw = ctx.encode(h, src, out=cw)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/5
    print(mode, "encode ms", t*1e3, "GB/s moved (k+n)*S*F:", (k+n)*S*F/t/1e9)
    if mode=="gather": ref=cw.clone()
print("equal:", torch.equal(ref,cw))
