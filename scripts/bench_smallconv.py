import torch, sys, time
sys.path.insert(0, '/root/repo')
from stylerenderer_amd.op.smallconv import _fwd, _dx, _dw
for (b,c,res) in ((16,128,256),(16,256,128),(16,512,64)):
    x=torch.randn(b,c,res,res,device='cuda'); ws=torch.randn(b,3,c,device='cuda'); g=torch.randn(b,3,res,res,device='cuda')
    for name,fn in (("fwd",lambda:_fwd(x,ws)),("dx",lambda:_dx(g,ws)),("dw",lambda:_dw(g,x))):
        fn(); torch.cuda.synchronize(); t=time.time()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt=(time.time()-t)/10
        print(f"{name} C{c} res{res}: {dt*1e6:.1f} us  {x.numel()*4/dt/1e12:.2f} TB/s (x bytes only)")
