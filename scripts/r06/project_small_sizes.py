# small ragged sizes through the f16 projection vs f64
import ctypes, sys, torch
sys.path.insert(0, "/root/repo")
from cleora_amd import _hip
L=_hip.lib(); dev=torch.device("cuda:0"); S=torch.cuda.current_stream().cuda_stream
g=torch.Generator(device=dev); g.manual_seed(1)
for n in (1, 63, 64, 65, 127, 128, 1000, 16384, 16385, 70001):
  for norm in (1,0,2):
    for scaled in (True, False):
        d=256
        x=torch.randn((n,d),generator=g,device=dev); x/=x.norm(dim=1,keepdim=True)
        bound=torch.ones(n,device=dev)
        rs=(2*torch.rand(n,generator=g,device=dev)-1).contiguous()
        mean=(torch.randn(d,generator=g,device=dev)*0.05).contiguous()
        t=torch.randn((d,d),generator=g,device=dev).contiguous()
        out=torch.full((n+1,d),7.0,device=dev)
        nd,fm=ctypes.c_int(0),ctypes.c_int(-1)
        _hip.check(L.cleora_project_bounded_dev(x.data_ptr(),d,n,d,mean.data_ptr(),t.data_ptr(),d,out.data_ptr(),d,rs.data_ptr() if scaled else None,bound.data_ptr() if scaled else None,norm,ctypes.byref(nd),ctypes.byref(fm),S))
        torch.cuda.synchronize()
        o=(x.double()-(rs.double()[:,None] if scaled else 1.0)*mean.double()[None,:])@t.double()
        if norm==1: o=o/o.norm(dim=1,keepdim=True)
        if norm==2: o=o/o.abs().sum(dim=1,keepdim=True)
        err=((out[:n].double()-o).norm(dim=1)/o.norm(dim=1)).max().item()
        guard=bool((out[n]==7.0).all())
        assert fm.value==1
        print(n,norm,scaled,"err %.2e"%err,"guard",guard, flush=True)
        assert err<2e-6 and guard
print("OK")
